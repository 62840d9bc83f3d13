set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_student_gpu.py -m gpu -q -k "fused_display" 2>&1 | tail -2
X="--cpu-seconds 0 --exact-frames 0 --full-frames 0 --batched-steps 0 --steps 1000 --warmup 100"
timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default fps', d['value'], d['roofline']['kernel_ms'], d['with_rgba8_d2h'])"
X="$X --d2h-frames 0"
for lib in ab_mfma ab_sin ab_fetch ab_barrier ab_zload ab_mfma_sin ab_fetch_barrier ab_all; do
  THA4_HIP_LIB=$R/build_variants/libtha4_$lib.so timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib fps', d['value'], d['roofline']['kernel_ms'])"
done 2>&1 | grep -v "^import\|^+" | tee gpurun_out/c13_ablate.txt
