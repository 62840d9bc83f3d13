# round 6, call 3: front16r_kernel (face + level 0 in registers, two 4-wave workgroups per CU) + batched taps in level1_16r: parity, A/B, ablations of the new forms
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_student_gpu.py tests/test_twin_gpu.py -m gpu -x -q > gpurun_out/c3_pytest.log 2>&1; tail -5 gpurun_out/c3_pytest.log
THA4_SWEEP_VARIANTS=default,frontregs0,fr664,fr1233,allregs0,ab_zload,ab_mfma,ab_fetch,ab_sin timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c3_sweep.txt 2>&1
cat gpurun_out/c3_sweep.txt
python bench.py --steps 200 --warmup 50 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/c3_bench.json; cut -c1-300 gpurun_out/c3_bench.json
python bench.py --batch 32 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/c3_bench_b32.json; cut -c1-300 gpurun_out/c3_bench_b32.json
