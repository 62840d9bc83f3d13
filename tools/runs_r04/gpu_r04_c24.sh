# round 4, call 24: per-lane tap table from four packed scalars (no vector loads from the argument block in front of the first operand request),
# norm_finalize_kernel's affine / FiLM operands requested with the moments: same-box A/B against the previous commit (libtha4_warm.so) + parity subset
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 3 warm=build_variants/libtha4_warm.so new=default 2>&1 | tee gpurun_out/c24_ab.txt
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py tests/test_twin_gpu.py -x -q -m gpu > gpurun_out/c24_pytest.log 2>&1; tail -2 gpurun_out/c24_pytest.log
