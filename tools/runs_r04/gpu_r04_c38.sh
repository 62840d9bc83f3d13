# round 4, call 38: host-computed launch constants (FastDiv reciprocals instead of ~8 run-time integer divisions per wave in the conv_small / conv_tile prologues):
# same-box A/B against the previous library (full model kernels of c29 = HEAD), then the reference-fixture parity tests of the full model
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 1 prev=build_variants/libtha4_c29.so new=default 2>&1 | tee gpurun_out/c38_ab.txt
timeout 115 python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "reference_fixture or plan" > gpurun_out/c38_pytest.log 2>&1; tail -2 gpurun_out/c38_pytest.log
