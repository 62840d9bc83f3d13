mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_full -- python $R/tools/time_full.py > $R/gpurun_out/pmc_full.log 2>&1
tail -2 $R/gpurun_out/pmc_full.log
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_full2 -- python $R/tools/time_full.py > $R/gpurun_out/pmc_full2.log 2>&1
ls $R/gpurun_out/pmc_full/*/ | head
