set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --cpu-seconds 0 --exact-frames 0 --full-frames 0 --batched-steps 0 --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default fps', d['value'], d['with_rgba8_d2h'])"
