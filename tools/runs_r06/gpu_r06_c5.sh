# round 6, call 5: A-fragment look-ahead of the register-resident kernels (ring of 3 register buffers), level 1 as two 4-wave workgroups per CU, tap batch sizes
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c5_pytest.log 2>&1; tail -2 gpurun_out/c5_pytest.log
THA4_SWEEP_VARIANTS=default,pf1,pf3,l1r423,tap3,tap6 timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c5_sweep.txt 2>&1
cat gpurun_out/c5_sweep.txt
THA4_SWEEP_VARIANTS=default,pf1 timeout 600 python tools/sweep.py run --steps 600 >> gpurun_out/c5_sweep.txt 2>&1
tail -2 gpurun_out/c5_sweep.txt
