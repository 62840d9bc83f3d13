# round 4, call 21: folded-norm table requested first (issue/finish split) + LDS-only barriers - A/B + parity
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 head=build_variants/libtha4_head.so new=default 2>&1 | tee gpurun_out/c21_ab.txt
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py tests/test_twin_gpu.py -x -q -m gpu > gpurun_out/c21_pytest.log 2>&1; tail -2 gpurun_out/c21_pytest.log
