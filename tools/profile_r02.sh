# Round-2 profiles (run on the GPU box): bash tools/profile_r02.sh   -> gpurun_out/{ps_*,pt_*,pf_*,pmcfull_*,bd_*}, then
# python tools/make_profile_md.py r02 (in the build container) turns them into profiles/r02_*.
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SB="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 0"
cd /tmp
# ---- student: kernel stats, two PMC passes, FETCH / WRITE passes ----
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_stats -- $SB > $R/gpurun_out/ps_stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/ps_pmc1 -- $SB > $R/gpurun_out/ps_pmc1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/ps_pmc3 -- $SB > $R/gpurun_out/ps_pmc3.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pt_fetch -- $SB > $R/gpurun_out/pt_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pt_write -- $SB > $R/gpurun_out/pt_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/ps_pmc1 gpurun_out/ps_pmc3 > gpurun_out/ps_pmc_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pt_fetch gpurun_out/pt_write > gpurun_out/pt_summary.txt 2>&1
python tools/traffic_json.py student gpurun_out/pt_fetch gpurun_out/pt_write --command "bench.py --steps 200 --warmup 50 --profile-frames 5 (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE)" -o gpurun_out/student_b1_traffic.json > /dev/null
cp $(ls gpurun_out/ps_stats/*/*kernel_stats.csv | head -1) gpurun_out/ps_kernel_stats.csv
rm -rf gpurun_out/ps_pmc1 gpurun_out/ps_pmc3 gpurun_out/ps_stats gpurun_out/pt_fetch gpurun_out/pt_write
# ---- full model: kernel stats (steady + cold), PMC, per-layer breakdown, traffic per steady / cold frame ----
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -- python $R/tools/time_full.py > $R/gpurun_out/pf_stats.log 2>&1
FB="python $R/tools/time_full.py --frames 3"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcfull_1 -- $FB > $R/gpurun_out/pmcfull_1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmcfull_2 -- $FB > $R/gpurun_out/pmcfull_2.log 2>&1
for m in steady cold; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pft_fetch_$m -- python $R/tools/time_full.py --frames 4 --mode $m > $R/gpurun_out/pft_fetch_$m.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pft_write_$m -- python $R/tools/time_full.py --frames 4 --mode $m > $R/gpurun_out/pft_write_$m.log 2>&1
done
THA4_DUMP_SCHEDULE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_full -- python $R/tools/time_full.py --frames 4 > $R/gpurun_out/bd_full.log 2> $R/gpurun_out/bd_full.err
cd $R
python tools/pmc_summary.py gpurun_out/pmcfull_1 gpurun_out/pmcfull_2 > gpurun_out/pmcfull_summary.txt 2>&1
python tools/traffic_json.py full gpurun_out/pft_fetch_steady gpurun_out/pft_write_steady --frames 4 --cold-fetch gpurun_out/pft_fetch_cold --cold-write gpurun_out/pft_write_cold --cold-frames 4 \
  --command "tools/time_full.py --frames 4 --mode steady|cold (+ 3 warm-up frames each)" -o gpurun_out/full_b1_traffic.json > /dev/null
grep "^conv " gpurun_out/bd_full.err > gpurun_out/bd_schedule.txt
python tools/conv_breakdown.py gpurun_out/bd_schedule.txt $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) gpurun_out/full_b1_layers.json > gpurun_out/bd_report.txt 2>&1
python tools/trace_gaps.py gpurun_out/bd_full 1200 > gpurun_out/bd_gaps.txt 2>&1
cp $(ls gpurun_out/pf_stats/*/*kernel_stats.csv | head -1) gpurun_out/pf_kernel_stats.csv
rm -rf gpurun_out/pf_stats gpurun_out/pmcfull_1 gpurun_out/pmcfull_2 gpurun_out/pft_* gpurun_out/bd_full
python tools/time_full.py > gpurun_out/pf_time.log 2>&1
tail -2 gpurun_out/pf_time.log; cat gpurun_out/student_b1_traffic.json | head -30; head -c 600 gpurun_out/full_b1_traffic.json; head -20 gpurun_out/ps_kernel_stats.csv
