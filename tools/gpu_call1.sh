# round-2 call 1: sanity (gpu tests), machine-model microbench, launch-gap trace of the full model, baseline bench
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1; tail -3 gpurun_out/c1_pytest.log
timeout 120 ./tools/microbench/mfma_valu_overlap > gpurun_out/c1_overlap.log 2>&1; cat gpurun_out/c1_overlap.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c1_trace -- python $R/tools/time_full.py --frames 6 > $R/gpurun_out/c1_trace.log 2>&1
cd $R
python tools/trace_gaps.py gpurun_out/c1_trace 2200 > gpurun_out/c1_gaps.txt 2>&1; cat gpurun_out/c1_gaps.txt
rm -rf gpurun_out/c1_trace
timeout 600 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; cat gpurun_out/c1_bench.json
