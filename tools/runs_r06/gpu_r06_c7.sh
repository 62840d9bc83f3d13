# round 6, call 7: hand-counted LDS fragment reads (inline-asm ds_read_b128 + s_waitcnt lgkmcnt(N)) against the compiler's waits (cwait = the previous build), stamps of the new form
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c7_pytest.log 2>&1; tail -2 gpurun_out/c7_pytest.log
THA4_SWEEP_VARIANTS=default,cwait,pf1 timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c7_sweep.txt 2>&1
THA4_SWEEP_VARIANTS=default,cwait timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c7_sweep.txt 2>&1
cat gpurun_out/c7_sweep.txt
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_stamps.so timeout 600 python tools/stamps_student.py > gpurun_out/c7_stamps.txt 2>&1
grep -v "chunk 1[123]" gpurun_out/c7_stamps.txt
