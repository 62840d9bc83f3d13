# HBM traffic counters of the student kernels, one counter family per pass (FETCH_SIZE and WRITE_SIZE in ONE pass aborts
# rocprofv3 on this image), plus a clean kernel-stats pass of the full model.  bash tools/profile_traffic.sh
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 100 --warmup 20 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pt_fetch -- $B > $R/gpurun_out/pt_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pt_write -- $B > $R/gpurun_out/pt_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -- python $R/tools/time_full.py > $R/gpurun_out/pf_stats.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pt_fetch gpurun_out/pt_write > gpurun_out/pt_summary.txt 2>&1
cp $(ls gpurun_out/pf_stats/*/*kernel_stats.csv | head -1) gpurun_out/pf_kernel_stats.csv
rm -rf gpurun_out/pt_fetch gpurun_out/pt_write gpurun_out/pf_stats
cat gpurun_out/pt_summary.txt; tail -2 gpurun_out/pf_stats.log; head -12 gpurun_out/pf_kernel_stats.csv
