set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c3_pytest.log 2>&1; tail -15 gpurun_out/c3_pytest.log
timeout 600 python bench.py > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; tail -3 gpurun_out/c3_bench.err; cat gpurun_out/c3_bench.json
timeout 600 python bench.py --model full > gpurun_out/c3_bench_full.json 2> gpurun_out/c3_bench_full.err; tail -3 gpurun_out/c3_bench_full.err; cat gpurun_out/c3_bench_full.json
timeout 300 python bench.py --batch 32 --steps 60 --warmup 10 --cpu-seconds 0 > gpurun_out/c3_bench_b32.json 2> gpurun_out/c3_bench_b32.err; tail -3 gpurun_out/c3_bench_b32.err; cat gpurun_out/c3_bench_b32.json
timeout 300 python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/c3_bench_fb8.json 2> gpurun_out/c3_bench_fb8.err; tail -3 gpurun_out/c3_bench_fb8.err; cat gpurun_out/c3_bench_fb8.json
