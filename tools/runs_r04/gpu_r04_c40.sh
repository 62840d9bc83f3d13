# round 4, call 40: the rest of tests/test_full_gpu.py + the remaining per-op device tests (as much as fits into the round's last GPU minute)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 50 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -q -m gpu -k "not (reference_fixture or plan or conv_kernels or fused_norm)" > gpurun_out/c40_pytest.log 2>&1; tail -3 gpurun_out/c40_pytest.log
