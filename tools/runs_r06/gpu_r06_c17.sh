# round 6, call 17: front16r_kernel with the level-0 waves at a higher issue priority than the face waves sharing their SIMDs (same-box A/B, two rounds)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
THA4_SWEEP_VARIANTS=default,l0prio1,l0prio2 timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c17_sweep.txt 2>&1
THA4_SWEEP_VARIANTS=default,l0prio1,l0prio2 timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c17_sweep.txt 2>&1
cat gpurun_out/c17_sweep.txt
