set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_ops_device.py tests/test_full_gpu.py -m gpu -x -q > gpurun_out/c6_pytest.log 2>&1; tail -3 gpurun_out/c6_pytest.log
python tools/time_full.py > gpurun_out/c6_time.log 2>&1; tail -2 gpurun_out/c6_time.log
THA4_SMALL_MAX_WGS=512 python tools/time_full.py > gpurun_out/c6_time_512.log 2>&1; tail -2 gpurun_out/c6_time_512.log
THA4_SMALL_1X1_MAX_PX=256 python tools/time_full.py > gpurun_out/c6_time_1x1_256.log 2>&1; tail -2 gpurun_out/c6_time_1x1_256.log
THA4_FUSED_NORM_MAX_TILES=16 python tools/time_full.py > gpurun_out/c6_time_f16.log 2>&1; tail -2 gpurun_out/c6_time_f16.log
THA4_FUSED_NORM_MAX_TILES=128 python tools/time_full.py > gpurun_out/c6_time_f128.log 2>&1; tail -2 gpurun_out/c6_time_f128.log
cd /tmp
THA4_DUMP_SCHEDULE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_full -- python $R/tools/time_full.py --frames 4 > $R/gpurun_out/bd_full.log 2> $R/gpurun_out/bd_full.err
cd $R
grep "^conv " gpurun_out/bd_full.err > gpurun_out/bd_schedule.txt
python tools/conv_breakdown.py gpurun_out/bd_schedule.txt $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) > gpurun_out/c6_bd.txt 2>&1
python tools/trace_gaps.py gpurun_out/bd_full 1500 > gpurun_out/c6_gaps.txt 2>&1
rm -rf gpurun_out/bd_full
head -45 gpurun_out/c6_bd.txt; head -30 gpurun_out/c6_gaps.txt
