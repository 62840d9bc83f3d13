# round 6, call 19: ablations of the final student kernels (results are wrong by construction): no MFMA / no sine / no weight fetch / no z taps / pose fold on 4 rows
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
THA4_SWEEP_VARIANTS=default,ab_posefold,ab_mfma,ab_fetch,ab_sin,ab_zload timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c19_ablations.txt 2>&1
THA4_SWEEP_VARIANTS=default,ab_posefold timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c19_ablations.txt 2>&1
cat gpurun_out/c19_ablations.txt
