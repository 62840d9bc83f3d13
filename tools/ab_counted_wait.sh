set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -m gpu -x -q > gpurun_out/c19_pytest.log 2>&1; tail -3 gpurun_out/c19_pytest.log
for rep in 1 2; do
for v in default nocount; do
  unset THA4_HIP_LIB
  if [ $v != default ]; then export THA4_HIP_LIB=$R/build_variants/libtha4_$v.so; fi
  python tools/time_full.py 2>/dev/null | sed "s/^/$v: /"
done
done
unset THA4_HIP_LIB
python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B8 default', d['value'])"
THA4_HIP_LIB=$R/build_variants/libtha4_nocount.so python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B8 nocount', d['value'])"
