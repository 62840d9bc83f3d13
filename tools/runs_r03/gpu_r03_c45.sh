# kernel-stats pass of the full model (batch 1) for the last commit of the round (the PMC / traffic passes stay those of gpu_r03_c41.sh)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -- python $R/tools/time_full.py > $R/gpurun_out/pf_stats.log 2>&1
cd $R
cp $(ls gpurun_out/pf_stats/*/*kernel_stats.csv | head -1) gpurun_out/pf_kernel_stats.csv
grep "full model" gpurun_out/pf_stats.log > gpurun_out/pf_time_profiled.log
rm -rf gpurun_out/pf_stats
head -5 gpurun_out/pf_kernel_stats.csv | cut -c1-120
