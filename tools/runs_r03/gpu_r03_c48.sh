mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 80 python bench.py --steps 20 --warmup 5 --cpu-seconds 2 > gpurun_out/c48_bench_driver.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c48_bench_driver.json')); print('driver-style', d['value'], d['ms_per_step'], d['full_model']['steady']['fps'], d['student_b32']['fps'], d['full_b8']['fps'], d['roofline']['frac'])"
