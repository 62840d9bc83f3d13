set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_student_gpu.py -m gpu -q -s -k "determinism_stress" 2>&1 | tail -3
X="--cpu-seconds 0 --d2h-frames 0 --exact-frames 0 --full-frames 0"
for lib in default pg1 hwsin_pg1 hwsin_stream; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_$lib.so; fi
  for r in 1 2; do timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib student fps', d['value'], d['roofline']['kernel_ms'])"; done
done 2>&1 | grep -v "^import\|^+" | tee gpurun_out/c5_l2_variants.txt
for lib in hwsin_pg1 hwsin_stream; do
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_$lib.so timeout 900 python -m pytest tests/test_student_gpu.py -m gpu -q -s -k "64_pose_sweep or determinism" 2>&1 | grep -E "PARITY sweep|passed|failed" | sed "s/^/$lib: /" | tee -a gpurun_out/c5_l2_variants.txt
done
