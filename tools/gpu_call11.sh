set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for env in "" "THA4_NO_SMALL_CONV=1" "THA4_FUSED_NORM_MAX_TILES=0"; do
  env $env python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B8 [$env]', d['value'], d['ms_per_step'])"
done
for env in "" "THA4_NO_SMALL_CONV=1"; do
  env $env python bench.py --model full --batch 2 --steps 30 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B2 [$env]', d['value'], d['ms_per_step'])"
done
