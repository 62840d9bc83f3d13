# round 4, call 37: finer prologue stamps of conv_small_kernel (wave start / scalar set-up / weights requested / per-lane set-up / loads issued ...)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 200 python tools/phase_timing_full.py --targets "tile=16x16 cin=512(cb 32) cout=512;tile=16x16 cin=256(cb 16) cout=256;tile=32x32 cin=256(cb 16) cout=256" 2>&1 | grep -v "^conv #" | tee gpurun_out/c37_phase.txt | cut -c1-220
