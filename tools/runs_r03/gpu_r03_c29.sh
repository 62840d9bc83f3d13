set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_student_gpu.py tests/test_image_io.py tests/test_full_gpu.py -m gpu -q -k "display or rgba8 or ingest" 2>&1 | tail -5
timeout 300 tools/microbench/pk_hazard 200 > gpurun_out/pk_hazard.txt 2>&1; echo rc=$?
grep -c FAIL gpurun_out/pk_hazard.txt; grep FAIL gpurun_out/pk_hazard.txt | head -80
