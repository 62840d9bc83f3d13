# PMC passes over the student bench (run on the GPU box): bash tools/pmc_student.sh <tag>
set -x
TAG=${1:-pmc}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 100 --warmup 20 --cpu-seconds 0 --profile-frames 5 --full-frames 0"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_1 -- $B > $R/gpurun_out/${TAG}_1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}_2 -- $B > $R/gpurun_out/${TAG}_2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/${TAG}_3 -- $B > $R/gpurun_out/${TAG}_3.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_1 gpurun_out/${TAG}_2 gpurun_out/${TAG}_3 > gpurun_out/${TAG}_summary.txt 2>&1
cat gpurun_out/${TAG}_summary.txt
