# round 4, call 11: the normalisation table's loads ahead of the weight / window loads in the prologues of conv_small / conv_tile / conv_point
# (same-box A/B against the previous commit's library), parity of the fused-norm paths
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 head=build_variants/libtha4_head.so new=default 2>&1 | tee gpurun_out/c11_ab.txt
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -x -q -m gpu > gpurun_out/c11_pytest.log 2>&1; tail -2 gpurun_out/c11_pytest.log
