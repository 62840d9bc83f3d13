set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c38_pytest.log 2>&1; tail -4 gpurun_out/c38_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c38_bench_driver.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c38_bench_driver.json')); print('driver-style', d['value'], d['ms_per_step'], d['full_model']['steady']['fps'], d['student_b32']['fps'], d['full_b8']['fps'])"
