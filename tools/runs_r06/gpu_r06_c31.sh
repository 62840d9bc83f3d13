# round 6, call 31: validation of the final tree (with the <8,2> tuning tile and its per-op case) (full GPU suite, smoke, default bench line)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/c31_pytest.log 2>&1; tail -3 gpurun_out/c31_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/c31_bench.json 2> gpurun_out/c31_bench.err; python -c "
import json
d=json.loads(open('gpurun_out/c31_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['repeats']['min'], d['repeats']['max'], r['kernel'], r['frac'], r['bound'], r.get('limiter'), r.get('mfma_busy'), r.get('kernel_rocprof_frac'))
print(d['full_model']['steady']['fps'], d['full_model']['cold']['fps'], d['student_b32']['fps'], d['full_b8']['fps'])
print(d['full_model']['roofline'].get('dominant_class_evidence'))"
