set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
export THA4_TUNING=1      # (added after the run: the plan knobs are only read with THA4_TUNING set; and the library needs -DTHA4_TILE16_BUILD)
for m in 0 1 2 4 8 16 7 15 31 0 7; do
  echo "== THA4_TILE16=$m"
  THA4_TILE16=$m timeout 300 python tools/time_full.py --frames 40 2>/dev/null | grep "full model"
  THA4_TILE16=$m timeout 300 python tools/time_full.py --batch 8 --frames 10 2>/dev/null | grep "full model"
done
THA4_TILE16=31 timeout 900 python -m pytest tests/test_full_gpu.py -q -x -k "fixture or plan or batch8" 2>&1 | tail -3
} 2>&1 | grep -v "^+" | tee gpurun_out/c37_tile16.txt
