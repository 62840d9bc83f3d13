# round 6, call 16: the tree with the "outer" mixed plan as default, write-through conv_tile outputs, unrequested leaf outputs not written, vectorised C16 writes of
# the image kernels: full GPU suite, then same-box A/Bs (write-through conv_point outputs; rounds 1-5's "write every output" through the tuning knob)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/c16_pytest.log 2>&1; tail -4 gpurun_out/c16_pytest.log
timeout 1200 python tools/ab_full.py --rounds 2 default=default pointwt=build_variants/libtha4_pointwt.so tilewt0=build_variants/libtha4_tilewt0.so writeall=default@THA4_TUNING=1,THA4_WRITE_ALL_OUTPUTS=1 > gpurun_out/c16_ab.txt 2>&1; cat gpurun_out/c16_ab.txt
