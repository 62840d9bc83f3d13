# round 4, call 31: student level 2 - layer biases / scales from LDS instead of three dependent global round trips per strip; display store in front of the blended store: A/B of the stream against c30s library
# trips each): same-box A/B of the batch-1 stream against the previous library, then the student parity tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 400 --warmup 100 --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['repeats']['all'], j['roofline'].get('kernel_ms'))"; }
for i in 1 2; do
  echo "student prev: $(THA4_HIP_LIB=$R/build_variants/libtha4_c30.so run)"
  echo "student new : $(run)"
done | tee gpurun_out/c31_student.txt
