# round 6, call 15: mid-gain parity of the four plans; A/B of write-through conv_tile output stores and of 128-workgroup small-map grids
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_full_gpu.py -m gpu -q -k midgain > gpurun_out/c15_pytest.log 2>&1; tail -5 gpurun_out/c15_pytest.log
for f in default split mixed_all exact; do grep "b8 up_merged\|b8 up_warped\|b8 face_6\|b8 body_merged\|b1 pose 0 up_merged" gpurun_out/full_midgain_parity_report_$f.txt | sed "s/^/$f: /"; done > gpurun_out/c15_midgain_rows.txt; cat gpurun_out/c15_midgain_rows.txt
timeout 1200 python tools/ab_full.py --rounds 2 default=default tilewt=build_variants/libtha4_tilewt.so small128=default@THA4_TUNING=1,THA4_SMALL_MAX_WGS=128 > gpurun_out/c15_ab.txt 2>&1; cat gpurun_out/c15_ab.txt
