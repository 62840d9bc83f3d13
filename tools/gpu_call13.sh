set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -m gpu -x -q > gpurun_out/c13_pytest.log 2>&1; tail -3 gpurun_out/c13_pytest.log
python tools/time_full.py > gpurun_out/c13_time.log 2>&1; tail -2 gpurun_out/c13_time.log
python tools/time_full.py > gpurun_out/c13_time2.log 2>&1; tail -2 gpurun_out/c13_time2.log
python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B8', d['value'], d['ms_per_step'])"
