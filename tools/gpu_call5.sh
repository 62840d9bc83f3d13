set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_ops_device.py tests/test_full_gpu.py -m gpu -x -q > gpurun_out/c5_pytest.log 2>&1; tail -5 gpurun_out/c5_pytest.log
python tools/time_full.py > gpurun_out/c5_time.log 2>&1; tail -2 gpurun_out/c5_time.log
THA4_FUSED_NORM_MAX_TILES=0 python tools/time_full.py > gpurun_out/c5_time_nofuse.log 2>&1; tail -2 gpurun_out/c5_time_nofuse.log
THA4_FUSED_NORM_MAX_TILES=16 python tools/time_full.py > gpurun_out/c5_time_fuse16.log 2>&1; tail -2 gpurun_out/c5_time_fuse16.log
THA4_SMALL_WANT_WGS=128 python tools/time_full.py > gpurun_out/c5_time_w128.log 2>&1; tail -2 gpurun_out/c5_time_w128.log
THA4_NO_SMALL_CONV=1 python tools/time_full.py > gpurun_out/c5_time_nosmall.log 2>&1; tail -2 gpurun_out/c5_time_nosmall.log
for mode in fused nofuse; do
  cd /tmp
  if [ $mode = nofuse ]; then export THA4_FUSED_NORM_MAX_TILES=0; fi
  THA4_DUMP_SCHEDULE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_full -- python $R/tools/time_full.py --frames 4 > $R/gpurun_out/bd_full.log 2> $R/gpurun_out/bd_full.err
  cd $R
  grep "^conv " gpurun_out/bd_full.err > gpurun_out/bd_schedule.txt
  python tools/conv_breakdown.py gpurun_out/bd_schedule.txt $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) > gpurun_out/c5_bd_$mode.txt 2>&1
  python tools/trace_gaps.py gpurun_out/bd_full 1500 > gpurun_out/c5_gaps_$mode.txt 2>&1
  rm -rf gpurun_out/bd_full
  head -60 gpurun_out/c5_bd_$mode.txt; head -14 gpurun_out/c5_gaps_$mode.txt
done
