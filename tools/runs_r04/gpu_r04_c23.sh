# round 4, call 23: warm_kernarg() - every line of the argument block touched at the top of each kernel (one miss latency instead of a chain of dependent ones):
# same-box A/B against HEAD (build_variants/libtha4_head.so), full model + student, then the parity subset
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 head=build_variants/libtha4_head.so new=default 2>&1 | tee gpurun_out/c23_ab.txt
run() { python bench.py --steps 400 --warmup 100 --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['repeats']['all'], j['roofline'].get('kernel_ms'))"; }
for i in 1 2; do
  echo "student head: $(THA4_HIP_LIB=$R/build_variants/libtha4_head.so run)"
  echo "student new : $(run)"
done | tee gpurun_out/c23_student.txt
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py tests/test_twin_gpu.py tests/test_student_gpu.py -x -q -m gpu > gpurun_out/c23_pytest.log 2>&1; tail -2 gpurun_out/c23_pytest.log
