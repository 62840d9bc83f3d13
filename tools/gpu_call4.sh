set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_ops_device.py tests/test_full_gpu.py -m gpu -x -q > gpurun_out/c4_pytest.log 2>&1; tail -15 gpurun_out/c4_pytest.log
python tools/time_full.py > gpurun_out/c4_time.log 2>&1; tail -2 gpurun_out/c4_time.log
THA4_NO_SMALL_CONV=1 python tools/time_full.py > gpurun_out/c4_time_nosmall.log 2>&1; tail -2 gpurun_out/c4_time_nosmall.log
THA4_SMALL_1X1_MAX_PX=0 python tools/time_full.py > gpurun_out/c4_time_no1x1.log 2>&1; tail -2 gpurun_out/c4_time_no1x1.log
THA4_SMALL_1X1_MAX_PX=300000 python tools/time_full.py > gpurun_out/c4_time_all1x1.log 2>&1; tail -2 gpurun_out/c4_time_all1x1.log
THA4_SMALL_WANT_WGS=128 python tools/time_full.py > gpurun_out/c4_time_w128.log 2>&1; tail -2 gpurun_out/c4_time_w128.log
THA4_SMALL_WANT_WGS=512 python tools/time_full.py > gpurun_out/c4_time_w512.log 2>&1; tail -2 gpurun_out/c4_time_w512.log
THA4_FUSED_NORM_MAX_TILES=0 python tools/time_full.py > gpurun_out/c4_time_nofuse.log 2>&1; tail -2 gpurun_out/c4_time_nofuse.log
THA4_FUSED_NORM_MAX_TILES=256 python tools/time_full.py > gpurun_out/c4_time_fuse256.log 2>&1; tail -2 gpurun_out/c4_time_fuse256.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4_trace -- python $R/tools/time_full.py --frames 6 > $R/gpurun_out/c4_trace.log 2>&1
cd $R
python tools/trace_gaps.py gpurun_out/c4_trace 1500 > gpurun_out/c4_gaps.txt 2>&1; cat gpurun_out/c4_gaps.txt
rm -rf gpurun_out/c4_trace
