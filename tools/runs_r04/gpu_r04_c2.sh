# round 4, call 2: conv_tile_kernel as four-wave workgroups, two per CU (NW = 4): TG_ID probe, parity of the fixture tests with every class
# switched, per-class A/B (THA4_TILE_NW4 mask: 1 <4,1>  2 <2,4>  4 <2,1>  8 <2,2>  16 <4,2>  32 <4,4>) and the start delay of the odd slot
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 60 tools/microbench/tg_id_probe > gpurun_out/c2_tg_probe.txt 2>&1; head -20 gpurun_out/c2_tg_probe.txt
THA4_TUNING=1 THA4_TILE_NW4=63 timeout 600 python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "fixture or batch" > gpurun_out/c2_pytest_nw4.log 2>&1; tail -3 gpurun_out/c2_pytest_nw4.log
T="THA4_TUNING=1"
python tools/ab_full.py default=default all63=default@$T,THA4_TILE_NW4=63 all63_d0=default@$T,THA4_TILE_NW4=63,THA4_TILE_DEPHASE=0 \
  all63_d3k=default@$T,THA4_TILE_NW4=63,THA4_TILE_DEPHASE=3000 all63_d12k=default@$T,THA4_TILE_NW4=63,THA4_TILE_DEPHASE=12000 \
  m1=default@$T,THA4_TILE_NW4=1 m2=default@$T,THA4_TILE_NW4=2 m4=default@$T,THA4_TILE_NW4=4 m8=default@$T,THA4_TILE_NW4=8 \
  m16=default@$T,THA4_TILE_NW4=16 m32=default@$T,THA4_TILE_NW4=32 default2=default 2>&1 | tee gpurun_out/c2_ab.txt
