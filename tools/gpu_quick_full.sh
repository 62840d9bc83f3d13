# quick check of a full-model kernel change (GPU box): parity tests of the full path, fps, per-kernel times
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -x -q -m gpu > gpurun_out/q_pytest.log 2>&1; tail -3 gpurun_out/q_pytest.log; fi
python tools/time_full.py --frames 60 2>/dev/null | grep "full model"
cd /tmp; rm -rf $R/gpurun_out/q_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_prof -- python $R/tools/time_full.py --mode steady --frames 20 > /dev/null 2>&1
cd $R
python tools/kernel_stats.py $(ls gpurun_out/q_prof/*/*kernel_trace.csv | head -1) > gpurun_out/q_stats.txt; cat gpurun_out/q_stats.txt
rm -rf gpurun_out/q_prof
