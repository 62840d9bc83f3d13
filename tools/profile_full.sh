set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
(timeout 900 python -m pytest tests/test_full_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r7_full.log
cat gpurun_out/full_parity_report.txt | head -40
cat gpurun_out/r7_full.log | tail -5
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_full -- python $R/tools/time_full.py > $R/gpurun_out/r7_prof.log 2>&1
tail -3 $R/gpurun_out/r7_prof.log
head -30 $R/gpurun_out/prof_full/*/*_kernel_stats.csv
