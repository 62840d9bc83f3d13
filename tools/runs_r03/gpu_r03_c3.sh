set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/sin_cliff.py 0 > gpurun_out/r03_sin_cliff.txt 2> gpurun_out/r03_sin_cliff.err; tail -14 gpurun_out/r03_sin_cliff.txt; tail -3 gpurun_out/r03_sin_cliff.err
timeout 600 python -m pytest tests/test_student_gpu.py -m gpu -q -s -k "hand_off" 2>&1 | grep -E "passed|failed|PARITY hand|Error" | tail
