# round 6, call 8: front16r_kernel on the ring with early barriers (one fragment pipeline per layer, LDS-DMA issue spread under the MFMAs, one-round-trip prologue)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c8_pytest.log 2>&1; tail -2 gpurun_out/c8_pytest.log
THA4_SWEEP_VARIANTS=default,cwait,fr6649,fr12384,spread0 timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c8_sweep.txt 2>&1
THA4_SWEEP_VARIANTS=default,cwait timeout 1500 python tools/sweep.py run --steps 600 >> gpurun_out/c8_sweep.txt 2>&1
cat gpurun_out/c8_sweep.txt
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_stamps.so timeout 600 python tools/stamps_student.py > gpurun_out/c8_stamps.txt 2>&1
grep -v "chunk 2[123]" gpurun_out/c8_stamps.txt | head -40
