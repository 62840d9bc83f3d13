# round 4, call 34: student streamed layers - scale + biases requested ahead of the GEMM (their wait no longer covers the next layers first weight chunk): A/B of the stream against c29s library
# trips each): same-box A/B of the batch-1 stream against the previous library, then the student parity tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 400 --warmup 100 --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['repeats']['all'], j['roofline'].get('kernel_ms'))"; }
for i in 1 2; do
  echo "student prev: $(THA4_HIP_LIB=$R/build_variants/libtha4_c29.so run)"
  echo "student new : $(run)"
done | tee gpurun_out/c34_student.txt
