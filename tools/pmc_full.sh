# PMC passes over a few full-model frames (run on the GPU box): bash tools/pmc_full.sh <tag>
set -x
TAG=${1:-pmcfull}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/tools/time_full.py --frames 3"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_1 -- $B > $R/gpurun_out/${TAG}_1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}_2 -- $B > $R/gpurun_out/${TAG}_2.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_1 gpurun_out/${TAG}_2 > gpurun_out/${TAG}_summary.txt 2>&1
rm -rf gpurun_out/${TAG}_1 gpurun_out/${TAG}_2
grep -E "^==|conv_tile|norm_fin|attention" gpurun_out/${TAG}_summary.txt
