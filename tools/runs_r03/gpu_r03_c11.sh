set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python tools/compare_libs.py default build_variants/libtha4_turns_wait0.so 32 2>&1 | tail -1 | tee gpurun_out/c11_compare.txt
timeout 900 python tools/compare_libs.py build_variants/libtha4_turns_pg1.so build_variants/libtha4_turns_pg1_wait0.so 32 2>&1 | tail -1 | tee -a gpurun_out/c11_compare.txt
X="--cpu-seconds 0 --d2h-frames 0 --exact-frames 0 --full-frames 0 --batched-steps 0"
for lib in default turns_pg1 poly; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$R/build_variants/libtha4_$lib.so; fi
  for r in 1 2; do timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib student fps', d['value'], d['roofline']['kernel_ms'])"; done
  timeout 900 python -m pytest tests/test_student_gpu.py -m gpu -q -s -k "64_pose_sweep or determinism or hand_off or output0_parity" 2>&1 | grep -E "PARITY sweep|PARITY hand|passed|failed" | sed "s/^/$lib: /"
done 2>&1 | grep -v "^import\|^+" | tee gpurun_out/c11_turns.txt
