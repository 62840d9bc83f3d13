# round 6, call 10: write-through z stores + prologue ahead of the ring burst: parity, same-box A/B (cwait = the previous build), spans
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c10_pytest.log 2>&1; tail -2 gpurun_out/c10_pytest.log
THA4_SWEEP_VARIANTS=default,cwait,zwt0 timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c10_sweep.txt 2>&1
THA4_SWEEP_VARIANTS=default,cwait,zwt0 timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c10_sweep.txt 2>&1
cat gpurun_out/c10_sweep.txt
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_stamps.so timeout 600 python tools/stamps_student.py > gpurun_out/c10_stamps.txt 2>&1
grep -v "chunk 2[0123]" gpurun_out/c10_stamps.txt | grep -A12 "workgroup 0 wave 0\|spans" | head -50
