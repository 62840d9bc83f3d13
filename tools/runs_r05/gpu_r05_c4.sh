#!/bin/bash
# round 5, call 4: norm_finalize with 8 loads in flight vs 4 (same-box A/B), the GPU suite on the library with ABI v5 (per-op timing), bench.py line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
timeout 500 python tools/ab_full.py --rounds 3 --no-b8 norm8=default norm4=build_variants/libtha4_norm4.so > gpurun_out/c4/ab.txt 2>&1
cat gpurun_out/c4/ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c4/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/c4/pytest.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c4/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['full_model']['steady']['fps'], d['full_model']['cold']['fps'], d['full_b8']['fps'], d['student_b32']['fps'])
print(json.dumps(d['full_model']['roofline'].get('dominant_class'))[:600])
print(d['full_model'].get('two_frames_in_flight'), d['full_model']['exact_fp32']['steady']['fps'])
PY
