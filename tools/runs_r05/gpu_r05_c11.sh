#!/bin/bash
# round 5, call 11: the complete GPU suite, smoke and the driver's bench command on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/c11/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/c11/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c11/smoke.txt 2>&1; tail -2 gpurun_out/c11/smoke.txt
SECONDS=0
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/c11/bench.json 2> gpurun_out/c11/bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c11/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['repeats']['median'], d['full_model']['steady']['fps'], d['full_model']['cold']['fps'], d['full_b8']['fps'], d['student_b32']['fps'], d['full_model']['exact_fp32']['steady']['fps'])
PY
