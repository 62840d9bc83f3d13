# round 6, call 29: mixed plan with ONLY the decomposer's first + down-sampling convolutions exact (its up-sampling convolutions and heads carry 3.3e-5 / 1.9e-5 of the
# split's error, fp64 attribution): parity on the mid-gain set and cost against the "outer" plan (same box)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
THA4_TUNING=1 THA4_DEC_DOWN_ONLY=1 timeout 900 python -m pytest tests/test_full_gpu.py -m gpu -q -k "midgain and default" > gpurun_out/c29_pytest.log 2>&1; tail -2 gpurun_out/c29_pytest.log
grep "b8 up_merged\|b8 up_warped\|b8 face_6\|b1 pose 0 up_merged" gpurun_out/full_midgain_parity_report_default.txt | sed "s/^/down-only: /" > gpurun_out/c29_rows.txt; cat gpurun_out/c29_rows.txt
timeout 1200 python tools/ab_full.py --rounds 2 outer=default downonly=default@THA4_TUNING=1,THA4_DEC_DOWN_ONLY=1 split=default@THA4_EXACT_DECOMPOSER=0 > gpurun_out/c29_ab.txt 2>&1; cat gpurun_out/c29_ab.txt
