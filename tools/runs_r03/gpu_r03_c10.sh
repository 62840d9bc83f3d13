set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/compare_libs.py default build_variants/libtha4_wait0.so 64 2>&1 | tail -1 | tee gpurun_out/c10_compare.txt
timeout 900 python tools/compare_libs.py build_variants/libtha4_hwsin_pg1.so build_variants/libtha4_hwsin_pg1_wait0.so 64 2>&1 | tail -1 | tee -a gpurun_out/c10_compare.txt
timeout 900 python tools/compare_libs.py build_variants/libtha4_hwsin.so build_variants/libtha4_hwsin_wait0.so 16 2>&1 | tail -1 | tee -a gpurun_out/c10_compare.txt
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/c10_pytest.log 2>&1; tail -5 gpurun_out/c10_pytest.log
timeout 900 python bench.py > gpurun_out/c10_bench.json 2> gpurun_out/c10_bench.err; cat gpurun_out/c10_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','settle_frames','with_rgba8_d2h','student_b32','full_b8','full_model')})"
tail -3 gpurun_out/c10_bench.err
