# round 6, call 24: level 1 with its first two tap batches requested in front of the prologue (parity, then same-box A/B against prologue-first), and the face workgroups of
# the front kernel warming level 1's prologue data + first weight chunks into every XCD's L2
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c24_pytest.log 2>&1; tail -3 gpurun_out/c24_pytest.log
rm -f gpurun_out/c24_sweep.txt
for i in 1 2; do THA4_SWEEP_VARIANTS=default,l1taps0,warm1 timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c24_sweep.txt 2>&1; done
cat gpurun_out/c24_sweep.txt
