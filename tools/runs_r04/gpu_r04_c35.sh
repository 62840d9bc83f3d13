# round 4, call 35: final checkpoint of the final library (c29's library + the student's bias-ahead change): whole GPU suite, smoke(), the driver-style bench lines,
# the student's rocprofv3 kernel-stats capture again
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/c35_pytest.log 2>&1; tail -3 gpurun_out/c35_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c35_bench_default.json 2> gpurun_out/c35_bench_default.err; head -c 700 gpurun_out/c35_bench_default.json; echo
timeout 300 python bench.py --model full --cpu-seconds 0 > gpurun_out/c35_bench_full.json 2>/dev/null; head -c 400 gpurun_out/c35_bench_full.json; echo
timeout 300 python bench.py --model full --batch 8 --cpu-seconds 0 > gpurun_out/c35_bench_full_b8.json 2>/dev/null; head -c 400 gpurun_out/c35_bench_full_b8.json; echo
cd /tmp
SB="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 0"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_stats -- $SB > $R/gpurun_out/ps_stats.log 2>&1
SB32="python $R/bench.py --batch 32 --characters lambda_00 --steps 24 --warmup 4 --cpu-seconds 0 --profile-frames 2 --settle-seconds 0 --repeats 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb32_stats -- $SB32 > $R/gpurun_out/pb32_stats.log 2>&1
cd $R
cp $(ls gpurun_out/ps_stats/*/*kernel_stats.csv | head -1) gpurun_out/ps_kernel_stats.csv
cp $(ls gpurun_out/pb32_stats/*/*kernel_stats.csv | head -1) gpurun_out/pb32_kernel_stats.csv
rm -rf gpurun_out/ps_stats gpurun_out/pb32_stats
python bench.py --batch 32 --characters lambda_00 --steps 64 --warmup 8 --cpu-seconds 0 --profile-frames 20 --repeats 0 > gpurun_out/pb32_bench.json 2>/dev/null
head -4 gpurun_out/ps_kernel_stats.csv | cut -c1-160; head -c 200 gpurun_out/pb32_bench.json
