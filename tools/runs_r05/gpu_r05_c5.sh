#!/bin/bash
# round 5, call 5: norm_finalize - tile-aware channel split and 8 loads in flight with a conditional tail - same-box A/B of the four combinations (batch 1), kernel stats of the new one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
timeout 700 python tools/ab_full.py --rounds 3 --no-b8 new=default old=build_variants/libtha4_norm4.so@THA4_TUNING=1,THA4_NORM_OLD_SPLIT=1 \
   split_only=build_variants/libtha4_norm4.so np8_only=default@THA4_TUNING=1,THA4_NORM_OLD_SPLIT=1 > gpurun_out/c5/ab.txt 2>&1
cat gpurun_out/c5/ab.txt
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5/stats -- python $R/tools/time_full.py > $R/gpurun_out/c5/stats.log 2>&1
cd $R
cp $(ls gpurun_out/c5/stats/*/*kernel_stats.csv | head -1) gpurun_out/c5/kernel_stats.csv; rm -rf gpurun_out/c5/stats
grep "norm_finalize\|affine_add\|gemv" gpurun_out/c5/kernel_stats.csv
timeout 600 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -x -q > gpurun_out/c5/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/c5/pytest.txt
