# round 6, call 23: level 2 with write-through stores of the posed frame (same-box A/B, two rounds); spread of "prologue done" over the front kernel's workgroups (stamps build)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/c23_sweep.txt
for i in 1 2; do THA4_SWEEP_VARIANTS=default,framewt timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c23_sweep.txt 2>&1; done
cat gpurun_out/c23_sweep.txt
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_stamps.so timeout 600 python tools/stamps_student.py > gpurun_out/c23_stamps.txt 2>&1
grep -A6 "spans" gpurun_out/c23_stamps.txt
