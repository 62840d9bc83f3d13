set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_student_gpu.py -m gpu -x -q > gpurun_out/c11_pytest.log 2>&1; tail -2 gpurun_out/c11_pytest.log
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_zwt0.so timeout 900 python -m pytest tests/test_student_gpu.py -m gpu -x -q -k "output0_parity" > gpurun_out/c11_pytest_zwt0.log 2>&1; tail -2 gpurun_out/c11_pytest_zwt0.log
