set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_full_gpu.py -m gpu -x -q > gpurun_out/c12_pytest.log 2>&1; tail -3 gpurun_out/c12_pytest.log
for b in 1 2 4 8; do
  python bench.py --model full --batch $b --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B$b', d['value'], d['ms_per_step'])"
done
THA4_FUSED_NORM_MAX_BATCH=8 python bench.py --model full --batch 4 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B4 fused', d['value'], d['ms_per_step'])"
THA4_FUSED_NORM_MAX_BATCH=0 python bench.py --model full --batch 2 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B2 nofuse', d['value'], d['ms_per_step'])"
