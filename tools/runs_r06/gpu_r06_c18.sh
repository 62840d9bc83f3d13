# round 6, call 18: is instruction fetch what the fully unrolled student kernels (front16r: 76 KB of straight-line code per wave) wait for?  I-cache counters.
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_LEVEL\|WAVE_DEP\|INSTS_BRANCH\|WAIT_IFETCH\)[A-Z_]*" | sort -u > $R/gpurun_out/c18_counters.txt; cat $R/gpurun_out/c18_counters.txt
B="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 0"
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pi1 -- $B > $R/gpurun_out/pi1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pi2 -- $B > $R/gpurun_out/pi2.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pi1 gpurun_out/pi2 2>&1 | grep "^==\|front16r\|level1_16r\|level2_16p" > gpurun_out/c18_icache.txt; cat gpurun_out/c18_icache.txt
tail -3 gpurun_out/pi1.log gpurun_out/pi2.log
rm -rf gpurun_out/pi1 gpurun_out/pi2
