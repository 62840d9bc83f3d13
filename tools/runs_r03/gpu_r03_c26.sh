set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
{
timeout 600 python tools/compare_libs.py build_variants/libtha4_turns_pg2_nopk.so build_variants/libtha4_turns_pg2_nopk_wait0.so 32 2>&1 | tail -1
timeout 600 python tools/compare_libs.py build_variants/libtha4_turns_pg2_noslp.so build_variants/libtha4_turns_pg2_noslp_wait0.so 32 2>&1 | tail -1
timeout 600 python tools/compare_libs.py build_variants/libtha4_nopk.so build_variants/libtha4_nopk_wait0.so 32 2>&1 | tail -1
timeout 600 python tools/compare_libs.py default build_variants/libtha4_nopk.so 32 2>&1 | tail -1
X="--cpu-seconds 0 --exact-frames 0 --batched-steps 0 --d2h-frames 0 --steps 1000 --warmup 100"
for lib in default nopk noslp turns_pg2_nopk; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$R/build_variants/libtha4_$lib.so; fi
  for r in 1 2; do timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib student fps', d['value'], d['roofline']['kernel_ms'], 'full', d['full_model']['steady']['fps'], d['full_model']['cold']['fps'])"; done
done
export THA4_HIP_LIB=$R/build_variants/libtha4_turns_pg2_nopk.so
timeout 900 python -m pytest tests/test_student_gpu.py -m gpu -q -s -k "64_pose_sweep or determinism" 2>&1 | grep -E "PARITY sweep|passed|failed" | sed "s/^/turns_pg2_nopk: /"
} 2>&1 | grep -v "^import\|^+" | tee gpurun_out/c26_nopk.txt
