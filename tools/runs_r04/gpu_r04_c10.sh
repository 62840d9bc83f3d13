# round 4, call 10: asymmetric issue priorities for the two co-resident four-wave workgroups of conv_tile_kernel<..., NW = 4>
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
T="THA4_TUNING=1"
python tools/ab_full.py default=default asym=default@$T,THA4_TILE_ASYM_PRIO=1 asym_all63=default@$T,THA4_TILE_ASYM_PRIO=1,THA4_TILE_NW4=63 \
  asym_min512=default@$T,THA4_TILE_ASYM_PRIO=1,THA4_TILE_NW4_MIN_WGS=512 asym_all63_min512=default@$T,THA4_TILE_ASYM_PRIO=1,THA4_TILE_NW4=63,THA4_TILE_NW4_MIN_WGS=512 \
  sym_min512=default@$T,THA4_TILE_NW4_MIN_WGS=512 default2=default 2>&1 | tee gpurun_out/c10_ab.txt
THA4_TUNING=1 THA4_TILE_ASYM_PRIO=1 timeout 600 python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "fixture or batch" > gpurun_out/c10_pytest.log 2>&1; tail -2 gpurun_out/c10_pytest.log
