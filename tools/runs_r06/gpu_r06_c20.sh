# round 6, call 20: level 2 with the weight copies' barrier behind the first strip's taps, level 1 with the one-round-trip prologue in front of its ring burst:
# parity (student + twin suites), then same-box A/B against the forms they replace
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c20_pytest.log 2>&1; tail -3 gpurun_out/c20_pytest.log
THA4_SWEEP_VARIANTS=default,l1pro0,l2early timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c20_sweep.txt 2>&1
THA4_SWEEP_VARIANTS=default,l1pro0,l2early timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c20_sweep.txt 2>&1
cat gpurun_out/c20_sweep.txt
