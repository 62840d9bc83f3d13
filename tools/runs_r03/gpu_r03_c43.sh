set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
{
for lib in prev default prev default; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$R/build_variants/libtha4_$lib.so; fi
  echo "== $lib"
  timeout 300 python tools/time_full.py --frames 40 2>/dev/null | grep "full model"
  timeout 300 python tools/time_full.py --batch 8 --frames 10 2>/dev/null | grep "full model"
done
unset THA4_HIP_LIB
timeout 900 python -m pytest tests/test_full_gpu.py -q -x -k "fixture or plan or batch8 or adversarial" 2>&1 | tail -2
} 2>&1 | grep -v "^+" | tee gpurun_out/c43_act.txt
