# round 4, call 29: final checkpoint on the final library - the whole GPU suite, smoke(), the driver-style bench lines, and the rocprofv3 kernel-stats
# captures of the four configurations (stats passes + the per-layer join only: the PMC / traffic passes of tools/profile_r04.sh were taken earlier this round)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/c29_pytest.log 2>&1; tail -3 gpurun_out/c29_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c29_bench_default.json 2> gpurun_out/c29_bench_default.err; head -c 900 gpurun_out/c29_bench_default.json; echo
timeout 300 python bench.py --model full --cpu-seconds 0 > gpurun_out/c29_bench_full.json 2>/dev/null; head -c 600 gpurun_out/c29_bench_full.json; echo
timeout 300 python bench.py --model full --batch 8 --cpu-seconds 0 > gpurun_out/c29_bench_full_b8.json 2>/dev/null; head -c 500 gpurun_out/c29_bench_full_b8.json; echo
cd /tmp
SB="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 0"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_stats -- $SB > $R/gpurun_out/ps_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -- python $R/tools/time_full.py > $R/gpurun_out/pf_stats.log 2>&1
THA4_DUMP_SCHEDULE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bd_full -- python $R/tools/time_full.py --frames 4 > $R/gpurun_out/bd_full.log 2> $R/gpurun_out/bd_full.err
SB32="python $R/bench.py --batch 32 --characters lambda_00 --steps 24 --warmup 4 --cpu-seconds 0 --profile-frames 2 --settle-seconds 0 --repeats 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb32_stats -- $SB32 > $R/gpurun_out/pb32_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pfb8_stats -- python $R/tools/time_full.py --batch 8 --frames 10 > $R/gpurun_out/pfb8_stats.log 2>&1
cd $R
cp $(ls gpurun_out/ps_stats/*/*kernel_stats.csv | head -1) gpurun_out/ps_kernel_stats.csv
cp $(ls gpurun_out/pf_stats/*/*kernel_stats.csv | head -1) gpurun_out/pf_kernel_stats.csv
cp $(ls gpurun_out/pb32_stats/*/*kernel_stats.csv | head -1) gpurun_out/pb32_kernel_stats.csv
cp $(ls gpurun_out/pfb8_stats/*/*kernel_stats.csv | head -1) gpurun_out/pfb8_kernel_stats.csv
grep "^conv " gpurun_out/bd_full.err > gpurun_out/bd_schedule.txt
python tools/conv_breakdown.py gpurun_out/bd_schedule.txt $(ls gpurun_out/bd_full/*/*kernel_trace.csv | head -1) > gpurun_out/bd_report.txt 2>&1
grep "full model batch" gpurun_out/pfb8_stats.log > gpurun_out/pfb8_time_profiled.log
grep "full model" gpurun_out/pf_stats.log > gpurun_out/pf_time_profiled.log
rm -rf gpurun_out/ps_stats gpurun_out/pf_stats gpurun_out/pb32_stats gpurun_out/pfb8_stats gpurun_out/bd_full
python tools/time_full.py > gpurun_out/pf_time.log 2>&1
python tools/time_full.py --batch 8 --frames 20 > gpurun_out/pfb8_time.log 2>&1
python bench.py --batch 32 --characters lambda_00 --steps 64 --warmup 8 --cpu-seconds 0 --profile-frames 20 --repeats 0 > gpurun_out/pb32_bench.json 2>/dev/null
tail -2 gpurun_out/pf_time.log; tail -1 gpurun_out/pfb8_time.log; head -5 gpurun_out/ps_kernel_stats.csv
