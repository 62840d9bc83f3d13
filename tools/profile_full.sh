# rocprofv3 kernel stats + PMC passes over the full-model timing script (run on the GPU box): bash tools/profile_full.sh
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -- python $R/tools/time_full.py > $R/gpurun_out/pf_stats.log 2>&1
B="python $R/tools/time_full.py --frames 3"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_pmc1 -- $B > $R/gpurun_out/pf_pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pf_pmc2 -- $B > $R/gpurun_out/pf_pmc2.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pf_pmc1 gpurun_out/pf_pmc2 > gpurun_out/pf_pmc_summary.txt 2>&1
cp $(ls gpurun_out/pf_stats/*/*kernel_stats.csv | head -1) gpurun_out/pf_kernel_stats.csv
rm -rf gpurun_out/pf_pmc1 gpurun_out/pf_pmc2 gpurun_out/pf_stats
python tools/time_full.py > gpurun_out/pf_time.log 2>&1
tail -2 gpurun_out/pf_time.log; tail -2 gpurun_out/pf_stats.log
head -40 gpurun_out/pf_kernel_stats.csv
cat gpurun_out/pf_pmc_summary.txt
