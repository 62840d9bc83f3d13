# rocprofv3 kernel stats + PMC passes over the student bench (run on the GPU box): bash tools/profile_student.sh
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_stats -- $B > $R/gpurun_out/ps_stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/ps_pmc1 -- $B > $R/gpurun_out/ps_pmc1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/ps_pmc3 -- $B > $R/gpurun_out/ps_pmc3.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/ps_pmc1 gpurun_out/ps_pmc3 > gpurun_out/ps_pmc_summary.txt 2>&1
cp $(ls gpurun_out/ps_stats/*/*kernel_stats.csv | head -1) gpurun_out/ps_kernel_stats.csv
rm -rf gpurun_out/ps_pmc1 gpurun_out/ps_pmc3 gpurun_out/ps_stats
timeout 300 python bench.py --steps 2000 --warmup 200 --full-frames 0 > gpurun_out/ps_bench.log 2>&1
tail -1 gpurun_out/ps_bench.log
cat gpurun_out/ps_kernel_stats.csv | head -12
cat gpurun_out/ps_pmc_summary.txt
