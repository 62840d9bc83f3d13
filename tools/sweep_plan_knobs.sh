# planner-knob sweep for the full model, batch 1 steady (GPU box): one line per setting in gpurun_out/knobs.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: > gpurun_out/knobs.txt
run() { echo "== $*" >> gpurun_out/knobs.txt; env THA4_TUNING=1 "$@" timeout 300 python tools/time_full.py --mode steady --frames 60 2>/dev/null | grep "full model" >> gpurun_out/knobs.txt; }
for k in "$@"; do run $k; done
cat gpurun_out/knobs.txt
