# round 3, GPU call 1: new parity tests, MFMA/VALU overlap (32x32x16), the v_sin_f32 cliff, baseline bench of this box
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 120 tools/microbench/mfma_valu_overlap2 > gpurun_out/r03_mfma_valu_overlap2.txt 2>&1; cat gpurun_out/r03_mfma_valu_overlap2.txt
timeout 600 python tools/sin_cliff.py 0 > gpurun_out/r03_sin_cliff.txt 2> gpurun_out/r03_sin_cliff.err; tail -12 gpurun_out/r03_sin_cliff.txt; tail -3 gpurun_out/r03_sin_cliff.err
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/c1_pytest.log 2>&1; grep -E "passed|failed|PARITY sweep|Error" gpurun_out/c1_pytest.log | tail -15
timeout 600 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; cat gpurun_out/c1_bench.json
