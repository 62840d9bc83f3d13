# round 4, call 39: per-op device tests of the convolution kernels with the host-computed launch constants (what fits into the round's last GPU minute)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 62 python -m pytest tests/test_ops_device.py -x -q -m gpu -k "conv_kernels or fused_norm" > gpurun_out/c39_pytest.log 2>&1; tail -2 gpurun_out/c39_pytest.log
