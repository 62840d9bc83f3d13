# round 6, call 25: the face workgroups of the front kernel warm level 1's prologue data + first weight chunks into every XCD's L2 (with the wave-end wait), same-box A/B
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/c25_sweep.txt
for i in 1 2; do THA4_SWEEP_VARIANTS=default,warm1 timeout 900 python tools/sweep.py run --steps 600 >> gpurun_out/c25_sweep.txt 2>&1; done
cat gpurun_out/c25_sweep.txt
