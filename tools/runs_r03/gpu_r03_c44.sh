# last validation of the round: twin comparison (full model), the whole GPU suite, smoke, the default bench line and the driver-style one
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/compare_libs.py default build_variants/libtha4_wait0.so 6 --full 2>&1 | tail -1 | tee gpurun_out/c44_compare.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/c44_pytest.log 2>&1; tail -3 gpurun_out/c44_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/c44_bench.json 2> gpurun_out/c44_bench.err; python -c "
import json; d=json.load(open('gpurun_out/c44_bench.json')); print({k: d[k] for k in ('value','ms_per_step','with_rgba8_d2h','student_b32','full_b8','full_model')}); print(d['roofline']['frac'], d['roofline']['kernel_ms'])"
