# per-layer breakdown of the full model (GPU box): usage  breakdown_full.sh <tag> [ENV=VALUE ...]
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
env "$@" THA4_DUMP_SCHEDULE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_$TAG -- python $R/tools/time_full.py > $R/gpurun_out/bd_$TAG.log 2> $R/gpurun_out/bd_$TAG.err
cd $R
grep "^conv " gpurun_out/bd_$TAG.err > gpurun_out/bd_${TAG}_schedule.txt
python tools/conv_breakdown.py gpurun_out/bd_${TAG}_schedule.txt $(ls gpurun_out/bd_$TAG/*/*kernel_trace.csv | head -1) > gpurun_out/bd_${TAG}_report.txt 2>&1
python tools/kernel_stats.py $(ls gpurun_out/bd_$TAG/*/*kernel_trace.csv | head -1) > gpurun_out/bd_${TAG}_kernels.txt 2>&1; rm -rf gpurun_out/bd_$TAG
tail -3 gpurun_out/bd_$TAG.log
