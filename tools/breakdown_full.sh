set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
THA4_DUMP_SCHEDULE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_full -- python $R/tools/time_full.py > $R/gpurun_out/bd_full.log 2> $R/gpurun_out/bd_full.err
cd $R
grep "^conv " gpurun_out/bd_full.err > gpurun_out/bd_schedule.txt
python tools/conv_breakdown.py gpurun_out/bd_schedule.txt $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) > gpurun_out/bd_report.txt 2>&1
python tools/kernel_stats.py $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) > gpurun_out/bd_kernels.txt 2>&1; rm -rf gpurun_out/bd_full
cat gpurun_out/bd_report.txt; tail -3 gpurun_out/bd_full.log
