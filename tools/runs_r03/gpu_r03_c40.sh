set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
{
X="--cpu-seconds 0 --exact-frames 0 --batched-steps 0 --d2h-frames 0 --full-frames 0 --steps 1000 --warmup 100"
for lib in prev default prev default; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$R/build_variants/libtha4_$lib.so; fi
  timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib student fps', d['value'], d['roofline']['kernel_ms'])"
done
unset THA4_HIP_LIB
timeout 600 python tools/compare_libs.py default build_variants/libtha4_wait0.so 32 2>&1 | tail -1
timeout 600 python -m pytest tests/test_student_gpu.py -q -x 2>&1 | tail -2
} 2>&1 | grep -v "^import\|^+" | tee gpurun_out/c40_fma.txt
