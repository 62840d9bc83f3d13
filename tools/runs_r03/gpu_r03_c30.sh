set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 tools/microbench/pk_hazard 200 > gpurun_out/pk_hazard2.txt 2>&1; echo rc=$?
grep -c FAIL gpurun_out/pk_hazard2.txt; grep FAIL gpurun_out/pk_hazard2.txt | grep -v "^mfma\|mfmaB" | head -120
