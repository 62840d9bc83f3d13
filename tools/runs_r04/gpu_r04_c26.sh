# round 4, call 26: conv_small_kernel epilogue without memory round trips (bias + activation codes requested with the first loads, stores last) and
# norm_finalize_kernel without indexed argument loads / conditional operand loads: same-box A/B against libtha4_varA.so (= previous state), in-kernel
# stamps of the small-map kernels with the warm argument block, parity subset
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 prev=build_variants/libtha4_varA.so new=default 2>&1 | tee gpurun_out/c26_ab.txt
timeout 300 python tools/phase_timing_full.py --targets "tile=16x16 cin=512(cb 32) cout=512;tile=32x32 cin=256(cb 16) cout=256;tile=16x16 cin=256(cb 16) cout=256;tile=128x128 cin=128(cb 8) cout=128;tile=256x256 cin=128(cb 8) cout=128" 2>&1 | grep -v "^conv #" | tee gpurun_out/c26_phase.txt | cut -c1-250
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py tests/test_twin_gpu.py -x -q -m gpu > gpurun_out/c26_pytest.log 2>&1; tail -2 gpurun_out/c26_pytest.log
