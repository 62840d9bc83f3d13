# round 6, call 32: the <8,2> tile on FOUR-wave workgroups (two per CU, THA4_TILE_TMB8=2): parity of the batch-8 plans, same-box A/B against <4,4> and the eight-wave <8,2>
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
THA4_TUNING=1 THA4_TILE_TMB8=2 timeout 900 python -m pytest tests/test_full_gpu.py -m gpu -q -k "batch8_plan or (midgain and default) or dense_batch" > gpurun_out/c32_pytest.log 2>&1; tail -2 gpurun_out/c32_pytest.log
timeout 1500 python tools/ab_full.py --rounds 3 --no-b1 default=default tmb8=default@THA4_TUNING=1,THA4_TILE_TMB8=1 tmb8nw4=default@THA4_TUNING=1,THA4_TILE_TMB8=2 > gpurun_out/c32_ab.txt 2>&1; cat gpurun_out/c32_ab.txt
