# final validation of the shipped build (no packed fp32, level 2 <12,64,1>) + the round's profile capture
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{ timeout 600 python tools/compare_libs.py default build_variants/libtha4_wait0.so 64 2>&1 | tail -1
  timeout 900 python tools/compare_libs.py default build_variants/libtha4_wait0.so 6 --full 2>&1 | tail -1; } | tee gpurun_out/c41_compare.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c41_pytest.log 2>&1; tail -5 gpurun_out/c41_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/c41_bench.json 2> gpurun_out/c41_bench.err; cat gpurun_out/c41_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','settle_frames','with_rgba8_d2h','student_b32','full_b8','full_model')}); print(d['roofline'])"
tail -3 gpurun_out/c41_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c41_bench_driver.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c41_bench_driver.json')); print('driver-style', d['value'], d['ms_per_step'])"
timeout 1200 bash tools/profile_r03.sh > gpurun_out/c41_profile.log 2>&1; tail -4 gpurun_out/c41_profile.log | cut -c1-400
