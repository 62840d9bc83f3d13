# round 4, call 19: conv_tile_kernel epilogue in three passes (all residual loads, then values, then all stores back to back) - same-box A/B + parity
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 head=build_variants/libtha4_head.so epi3=default 2>&1 | tee gpurun_out/c19_ab.txt
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py tests/test_twin_gpu.py -x -q -m gpu > gpurun_out/c19_pytest.log 2>&1; tail -2 gpurun_out/c19_pytest.log
