#!/bin/bash
# round 5, call 9: wall time of the driver's bench command on the final tree; the multi-rank RCCL test worker at world 1 (a syntax / logic smoke of the code 1-GPU leases skip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
T0=$(date +%s.%N)
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/c9/bench.json 2> gpurun_out/c9/bench.err; echo "bench rc=$?"
T1=$(date +%s.%N)
echo "bench wall seconds: $(echo "$T1 - $T0" | bc)" | tee gpurun_out/c9/bench_time.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c9/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['full_model']['steady']['fps'], d['full_model']['two_frames_in_flight'], d['full_model']['exact_fp32']['steady']['fps'])
PY
timeout 300 python tools/runs_r05/worker_world1.py > gpurun_out/c9/worker_world1.txt 2>&1
tail -4 gpurun_out/c9/worker_world1.txt
