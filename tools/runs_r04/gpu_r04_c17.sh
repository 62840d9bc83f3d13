# round 4, call 17: final checkpoint - the whole GPU suite, smoke(), the driver-style bench lines
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/c17_pytest.log 2>&1; tail -3 gpurun_out/c17_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c17_bench_default.json 2> gpurun_out/c17_bench_default.err; head -c 700 gpurun_out/c17_bench_default.json; echo
timeout 300 python bench.py --model full --cpu-seconds 0 > gpurun_out/c17_bench_full.json 2>/dev/null; head -c 500 gpurun_out/c17_bench_full.json; echo
timeout 300 python bench.py --model full --batch 8 --cpu-seconds 0 > gpurun_out/c17_bench_full_b8.json 2>/dev/null; head -c 400 gpurun_out/c17_bench_full_b8.json; echo
