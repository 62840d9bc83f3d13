# round 4, call 14: conv_tile_kernel requests its first window before it builds the folded normalisation table (same-box A/B) + parity
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 3 --no-b8 head=build_variants/libtha4_head.so early=default 2>&1 | tee gpurun_out/c14_ab.txt
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -x -q -m gpu > gpurun_out/c14_pytest.log 2>&1; tail -2 gpurun_out/c14_pytest.log
