# Round-6 profiles (run on the GPU box): bash tools/profile_r06.sh - the captures of rounds 2-5 (tools/profile_r03.sh: student batch 1 / batch 32, full model
# batch 1 / batch 8: rocprofv3 kernel stats, SQ PMC passes, FETCH_SIZE / WRITE_SIZE passes, per-layer breakdown) on this round's FINAL library, plus the
# machine-readable SQ summary of the student kernels bench.py quotes as roofline.mfma_busy / .limiter (tools/pmc_json.py: it needs the raw counter
# directories, which profile_r02.sh deletes after summarising, so its three passes run here first).
# python tools/make_profile_md.py r06 (build container) then turns gpurun_out/ into profiles/r06_*.
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for B in 1 32; do
  if [ $B = 1 ]; then SB="python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 0"
  else SB="python $R/bench.py --batch 32 --characters lambda_00 --steps 24 --warmup 4 --cpu-seconds 0 --profile-frames 2 --settle-seconds 0 --repeats 0"; fi
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pj${B}_a -- $SB > $R/gpurun_out/pj${B}_a.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/pj${B}_b -- $SB > $R/gpurun_out/pj${B}_b.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pj${B}_c -- $SB > $R/gpurun_out/pj${B}_c.log 2>&1
  (cd $R && python tools/pmc_json.py gpurun_out/pj${B}_a gpurun_out/pj${B}_b gpurun_out/pj${B}_c --batch $B -o gpurun_out/student_b${B}_pmc.json; rm -rf gpurun_out/pj${B}_a gpurun_out/pj${B}_b gpurun_out/pj${B}_c)
done
# the same three SQ passes for the full model (batch 1 steady + cold frames, batch 8): per kernel template instance (tools/pmc_json.py --mode full)
for B in 1 8; do
  if [ $B = 1 ]; then FB="python $R/tools/time_full.py --frames 3"; else FB="python $R/tools/time_full.py --batch 8 --frames 5"; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pjf${B}_a -- $FB > $R/gpurun_out/pjf${B}_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/pjf${B}_b -- $FB > $R/gpurun_out/pjf${B}_b.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pjf${B}_c -- $FB > $R/gpurun_out/pjf${B}_c.log 2>&1
  (cd $R && python tools/pmc_json.py --mode full gpurun_out/pjf${B}_a gpurun_out/pjf${B}_b gpurun_out/pjf${B}_c --batch $B -o gpurun_out/full_b${B}_pmc.json > gpurun_out/pjf${B}_summary.txt; rm -rf gpurun_out/pjf${B}_a gpurun_out/pjf${B}_b gpurun_out/pjf${B}_c)
done
cd $R
cat gpurun_out/student_b1_pmc.json
bash tools/profile_r03.sh
