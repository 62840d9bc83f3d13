# round 6, call 4: rocprofv3 kernel stats + SQ PMC passes of the student stream on the register-resident kernels (what do the waves wait for?)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0 --repeats 0"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_stats -- $B > $R/gpurun_out/ps_stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/ps_pmc1 -- $B > $R/gpurun_out/ps_pmc1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/ps_pmc3 -- $B > $R/gpurun_out/ps_pmc3.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/ps_pmc4 -- $B > $R/gpurun_out/ps_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/ps_pmc1 gpurun_out/ps_pmc3 gpurun_out/ps_pmc4 > gpurun_out/c4_pmc_summary.txt 2>&1
cp $(ls gpurun_out/ps_stats/*/*kernel_stats.csv | head -1) gpurun_out/c4_kernel_stats.csv
rm -rf gpurun_out/ps_pmc1 gpurun_out/ps_pmc3 gpurun_out/ps_pmc4 gpurun_out/ps_stats
head -8 gpurun_out/c4_kernel_stats.csv
cat gpurun_out/c4_pmc_summary.txt
tail -3 gpurun_out/ps_pmc4.log
