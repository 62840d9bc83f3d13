# round 6, call 30: the <8,2> output tile of conv_tile_kernel (round-5 review, task 2) as a tuning option: parity of the batch-8 plans with it, same-box A/B at batch 8,
# SQ counters of the class (matrix pipe busy, VALU per MFMA) against <4,4>
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
THA4_TUNING=1 THA4_TILE_TMB8=1 timeout 900 python -m pytest tests/test_full_gpu.py -m gpu -q -k "batch8_plan or (midgain and default) or dense_batch" > gpurun_out/c30_pytest.log 2>&1; tail -2 gpurun_out/c30_pytest.log
grep "b8 up_merged" gpurun_out/full_midgain_parity_report_default.txt
timeout 1200 python tools/ab_full.py --rounds 3 --no-b1 default=default tmb8=default@THA4_TUNING=1,THA4_TILE_TMB8=1 > gpurun_out/c30_ab.txt 2>&1; cat gpurun_out/c30_ab.txt
cd /tmp
FB="python $R/tools/time_full.py --batch 8 --frames 5"
THA4_TUNING=1 THA4_TILE_TMB8=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/p8_a -- $FB > $R/gpurun_out/p8_a.log 2>&1
THA4_TUNING=1 THA4_TILE_TMB8=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/p8_b -- $FB > $R/gpurun_out/p8_b.log 2>&1
THA4_TUNING=1 THA4_TILE_TMB8=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p8_c -- $FB > $R/gpurun_out/p8_c.log 2>&1
cd $R
python tools/pmc_json.py --mode full gpurun_out/p8_a gpurun_out/p8_b gpurun_out/p8_c --batch 8 -o gpurun_out/c30_full_b8_tmb8_pmc.json > gpurun_out/c30_pmc_summary.txt 2>&1
rm -rf gpurun_out/p8_a gpurun_out/p8_b gpurun_out/p8_c
python - <<'PY'
import json
d=json.load(open('gpurun_out/c30_full_b8_tmb8_pmc.json'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1].get('launches_per_pass',0)*kv[1].get('avg_us',0))[:8]:
    print(f"{k[:40]:40s} n={v.get('launches_per_pass')} avg_us={v.get('avg_us')} busy={v.get('mfma_busy')} valu/mfma={v.get('valu_per_mfma')} stall={v.get('issue_stall')} parked={v.get('parked')}")
PY
