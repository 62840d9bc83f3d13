#!/bin/bash
# round 5, call 7: moment accumulators also where the per-tile route would need 3-8 load rounds (17-64 tiles): A/B of the threshold, then the GPU suite + bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
timeout 700 python tools/ab_full.py --rounds 3 --no-b8 acc16=default acc64=default@THA4_TUNING=1,THA4_ACC_MIN_TILES=64 finalize=default@THA4_TUNING=1,THA4_NO_MOMENT_ACC=1 > gpurun_out/c7/ab.txt 2>&1
cat gpurun_out/c7/ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c7/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/c7/pytest.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c7/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['full_model']['steady']['fps'], d['full_model']['cold']['fps'], d['full_b8']['fps'], d['student_b32']['fps'])
PY
