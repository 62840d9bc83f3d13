set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B > $R/gpurun_out/r4_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/prof_pmc1 -- $B > $R/gpurun_out/r4_pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc2 -- $B > $R/gpurun_out/r4_pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/prof_pmc3 -- $B > $R/gpurun_out/r4_pmc3.log 2>&1
cd $R
timeout 300 python bench.py --steps 2000 --warmup 200 > gpurun_out/r4_bench.log 2>&1
tail -2 gpurun_out/r4_bench.log
ls gpurun_out/prof_stats/*/ gpurun_out/prof_pmc1/*/ | head -20
tail -3 gpurun_out/r4_pmc1.log
