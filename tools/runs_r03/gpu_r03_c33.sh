set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
{
X="--cpu-seconds 0 --exact-frames 0 --batched-steps 0 --d2h-frames 0 --full-frames 0 --steps 1000 --warmup 100"
for lib in default l1w4 l1w8x2 l1hb2 default l1w4 l1w8x2; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$R/build_variants/libtha4_$lib.so; fi
  timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib student fps', d['value'], d['roofline']['kernel_ms'])"
done
unset THA4_HIP_LIB
timeout 600 python tools/compare_libs.py build_variants/libtha4_l1w4.so build_variants/libtha4_l1w4_wait0.so 16 2>&1 | tail -1
timeout 600 python tools/compare_libs.py build_variants/libtha4_l1w8x2.so build_variants/libtha4_l1w8x2_wait0.so 16 2>&1 | tail -1
} 2>&1 | grep -v "^import\|^+" | tee gpurun_out/c33_l1.txt
