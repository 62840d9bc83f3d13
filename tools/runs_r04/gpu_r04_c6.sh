# round 4, call 6: DPP statistics in the convolution epilogues (default library), NW = 4 + phase priorities with longer start delays of
# the odd slot, the phased tap order; parity subset with the candidate configuration
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
T="THA4_TUNING=1"; P=build_variants/libtha4_prio.so; Q=build_variants/libtha4_prio_phased.so
python tools/ab_full.py default=default prio_m50=$P@$T,THA4_TILE_NW4=50 prio_m50_d20k=$P@$T,THA4_TILE_NW4=50,THA4_TILE_DEPHASE=20000 \
  prio_m50_d40k=$P@$T,THA4_TILE_NW4=50,THA4_TILE_DEPHASE=40000 prio_m51_d20k=$P@$T,THA4_TILE_NW4=51,THA4_TILE_DEPHASE=20000 \
  phased=$Q phased_m50=$Q@$T,THA4_TILE_NW4=50 default2=default 2>&1 | tee gpurun_out/c6_ab.txt
THA4_HIP_LIB=$R/$P THA4_TUNING=1 THA4_TILE_NW4=50 timeout 600 python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "fixture or batch or determinism or poison" > gpurun_out/c6_pytest.log 2>&1; tail -3 gpurun_out/c6_pytest.log
timeout 600 python -m pytest tests/test_full_gpu.py tests/test_ops_device.py -x -q -m gpu > gpurun_out/c6_pytest_default.log 2>&1; tail -3 gpurun_out/c6_pytest_default.log
