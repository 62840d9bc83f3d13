set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_student_gpu.py -m gpu -x -q > gpurun_out/c7_pytest.log 2>&1; tail -3 gpurun_out/c7_pytest.log
timeout 600 python bench.py --cpu-seconds 0 --full-frames 0 > gpurun_out/c7_bench.json 2> gpurun_out/c7_bench.err; cat gpurun_out/c7_bench.json
timeout 600 python bench.py --cpu-seconds 0 --full-frames 0 --batch 32 --steps 60 --warmup 10 > gpurun_out/c7_bench_b32.json 2> gpurun_out/c7_bench_b32.err; cat gpurun_out/c7_bench_b32.json
