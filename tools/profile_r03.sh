# Round-3 profiles (run on the GPU box): bash tools/profile_r03.sh  -> gpurun_out/{ps_*,pt_*,pf_*,pmcfull_*,bd_*,pb32_*,pfb8_*}, then
# python tools/make_profile_md.py r03 (in the build container) turns them into profiles/r03_*.
# Batch-1 captures as in round 2 (tools/profile_r02.sh) + the batched configurations BASELINE.json names: configs[3] one GPU's share
# (student, batch 32) and configs[4] one GPU's share (full model, batch 8), each with kernel stats, SQ PMC pass and FETCH / WRITE passes.
set -x
bash tools/profile_r02.sh > gpurun_out/profile_r02_part.log 2>&1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES"      # (round 5: LDS counters of the batched captures)
cd /tmp
# ---- student, batch 32 ----
SB="python $R/bench.py --batch 32 --characters lambda_00 --steps 24 --warmup 4 --cpu-seconds 0 --profile-frames 2 --settle-seconds 0 --repeats 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb32_stats -- $SB > $R/gpurun_out/pb32_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc $PMC1 --output-format csv -d $R/gpurun_out/pb32_pmc1 -- $SB > $R/gpurun_out/pb32_pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pb32_fetch -- $SB > $R/gpurun_out/pb32_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pb32_write -- $SB > $R/gpurun_out/pb32_write.log 2>&1
# ---- full model, batch 8 ----
FB="python $R/tools/time_full.py --batch 8 --frames 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pfb8_stats -- python $R/tools/time_full.py --batch 8 --frames 10 > $R/gpurun_out/pfb8_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $PMC1 --output-format csv -d $R/gpurun_out/pfb8_pmc1 -- $FB > $R/gpurun_out/pfb8_pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d $R/gpurun_out/pfb8_pmc2 -- $FB > $R/gpurun_out/pfb8_pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pfb8_fetch -- $FB > $R/gpurun_out/pfb8_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pfb8_write -- $FB > $R/gpurun_out/pfb8_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pb32_pmc1 > gpurun_out/pb32_pmc_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pb32_fetch gpurun_out/pb32_write > gpurun_out/pb32_traffic_summary.txt 2>&1
python tools/traffic_json.py student gpurun_out/pb32_fetch gpurun_out/pb32_write --batch 32 --command "bench.py --batch 32 --characters lambda_00 --steps 24 --warmup 4 (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE)" -o gpurun_out/student_b32_traffic.json > /dev/null
python tools/pmc_summary.py gpurun_out/pfb8_pmc1 gpurun_out/pfb8_pmc2 > gpurun_out/pfb8_pmc_summary.txt 2>&1
python tools/traffic_json.py full gpurun_out/pfb8_fetch gpurun_out/pfb8_write --batch 8 --steps 8 --command "tools/time_full.py --batch 8 --frames 5 (+ 3 warm-up steps; rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE)" -o gpurun_out/full_b8_traffic.json > /dev/null
cp $(ls gpurun_out/pb32_stats/*/*kernel_stats.csv | head -1) gpurun_out/pb32_kernel_stats.csv
cp $(ls gpurun_out/pfb8_stats/*/*kernel_stats.csv | head -1) gpurun_out/pfb8_kernel_stats.csv
grep "full model batch" gpurun_out/pfb8_stats.log > gpurun_out/pfb8_time_profiled.log
rm -rf gpurun_out/pb32_stats gpurun_out/pb32_pmc1 gpurun_out/pb32_fetch gpurun_out/pb32_write gpurun_out/pfb8_stats gpurun_out/pfb8_pmc1 gpurun_out/pfb8_pmc2 gpurun_out/pfb8_fetch gpurun_out/pfb8_write
python tools/time_full.py --batch 8 --frames 20 > gpurun_out/pfb8_time.log 2>&1
python bench.py --batch 32 --characters lambda_00 --steps 64 --warmup 8 --cpu-seconds 0 --profile-frames 20 --repeats 0 > gpurun_out/pb32_bench.json 2>/dev/null
tail -1 gpurun_out/pfb8_time.log; head -c 400 gpurun_out/pb32_bench.json; head -c 500 gpurun_out/student_b32_traffic.json; head -c 500 gpurun_out/full_b8_traffic.json
