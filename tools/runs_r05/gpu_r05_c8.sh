#!/bin/bash
# round 5, call 8: the final library - GPU suite, smoke, the driver's bench line and the three other configurations' lines (raw records for profiles/r05_raw)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/c8/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/c8/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c8/smoke.txt 2>&1; tail -2 gpurun_out/c8/smoke.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/c8/bench_student_b1.json 2> gpurun_out/c8/bench_student_b1.err; echo "bench rc=$?"
timeout 300 python bench.py --model full > gpurun_out/c8/bench_full_b1.json 2>/dev/null
timeout 300 python bench.py --batch 32 --cpu-seconds 0 > gpurun_out/c8/bench_student_b32.json 2>/dev/null
timeout 300 python bench.py --model full --batch 8 --cpu-seconds 0 > gpurun_out/c8/bench_full_b8.json 2>/dev/null
python - <<'PY'
import json
for f in ("student_b1","full_b1","student_b32","full_b8"):
    try:
        d=json.loads(open(f'gpurun_out/c8/bench_{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d.get('repeats',{}).get('median'), d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
d=json.loads(open('gpurun_out/c8/bench_student_b1.json').read().strip().splitlines()[-1])
print(d['full_model']['steady']['fps'], d['full_model']['cold']['fps'], d['full_b8']['fps'], d['student_b32']['fps'], d['full_model']['exact_fp32']['steady']['fps'], d['full_model']['two_frames_in_flight'].get('fps'))
print(d['roofline']['frac'], d['roofline']['kernel_ms'])
PY
