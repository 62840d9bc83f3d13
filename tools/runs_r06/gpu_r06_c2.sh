# round 6, call 2: level1_16r_kernel (activations in registers) - parity on the device, then same-box A/B of its geometries against the old kernel
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c2_pytest.log 2>&1; tail -5 gpurun_out/c2_pytest.log
rm -f build_variants/libtha4_default.so
THA4_SWEEP_VARIANTS=default,l1regs0,l1r816,l1r413,l1r824 timeout 1200 python tools/sweep.py run --steps 600 > gpurun_out/c2_l1r_sweep.txt 2>&1
cat gpurun_out/c2_l1r_sweep.txt
python bench.py --steps 200 --warmup 50 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/c2_bench.json; cut -c1-400 gpurun_out/c2_bench.json
