#!/bin/bash
# round 5, call 3: pixel <-> column permutation (conflict-free window reads): same-box A/B against the identity build, LDS conflict counters of both at batch 8, GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
timeout 700 python tools/ab_full.py --rounds 3 perm=default identity=build_variants/libtha4_identity_pixels.so > gpurun_out/c3/ab.txt 2>&1
cat gpurun_out/c3/ab.txt
R=$GRAFT_REPO_ROOT
cd /tmp
for v in perm identity; do
  if [ $v = identity ]; then export THA4_HIP_LIB=$R/build_variants/libtha4_identity_pixels.so; else unset THA4_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/c3/pmc_$v -- python $R/tools/time_full.py --batch 8 --frames 3 > $R/gpurun_out/c3/pmc_$v.log 2>&1
done
unset THA4_HIP_LIB
cd $R
for v in perm identity; do python tools/pmc_summary.py gpurun_out/c3/pmc_$v > gpurun_out/c3/pmc_${v}_summary.txt 2>&1; rm -rf gpurun_out/c3/pmc_$v; done
grep -E "conv_tile_kernel<4, 4, 0, 1, 4>|conv_tile_kernel<4, 2, 0, 1, 4>|conv_tile_kernel<4, 1, 0" gpurun_out/c3/pmc_*_summary.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c3/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/c3/pytest.txt
