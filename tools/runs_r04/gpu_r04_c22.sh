# round 4, call 22: where do kernel arguments live?  In-kernel stamps put 1.4-3.2 us between a workgroup's entry and its first vector-memory request
# (scalar loads from the argument block).  A/B of the runtime's HIP_FORCE_DEV_KERNARG (argument blocks in device memory vs host memory), full model + student.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 plain=default dev=default@HIP_FORCE_DEV_KERNARG=1 host=default@HIP_FORCE_DEV_KERNARG=0 2>&1 | tee gpurun_out/c22_ab.txt
run() { python bench.py --steps 400 --warmup 100 --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['repeats'])"; }
for i in 1 2; do
  echo "student plain: $(run)"
  echo "student dev  : $(HIP_FORCE_DEV_KERNARG=1 run)"
  echo "student host : $(HIP_FORCE_DEV_KERNARG=0 run)"
done | tee gpurun_out/c22_student.txt
