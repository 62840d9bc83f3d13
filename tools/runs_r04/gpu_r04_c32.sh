# round 4, call 32: per-layer join of the schedule dump with a kernel trace again (tools/conv_breakdown.py did not know the merged parity classes: c29's table was misaligned)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
THA4_DUMP_SCHEDULE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_full -- python $R/tools/time_full.py --frames 4 > $R/gpurun_out/bd_full.log 2> $R/gpurun_out/bd_full.err
cd $R
grep "^conv " gpurun_out/bd_full.err > gpurun_out/bd_schedule.txt
python tools/conv_breakdown.py gpurun_out/bd_schedule.txt $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) > gpurun_out/bd_report.txt 2>&1
python tools/trace_gaps.py gpurun_out/bd_full 1200 > gpurun_out/bd_gaps.txt 2>&1
rm -rf gpurun_out/bd_full
head -8 gpurun_out/bd_report.txt | cut -c1-200; head -12 gpurun_out/bd_gaps.txt | cut -c1-200
