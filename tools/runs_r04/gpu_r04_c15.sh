# round 4, call 15: batch 8 - smaller tiles / more workgroups per CU through the planner's existing knobs (THA4_WANT_WGS) x four-wave classes
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
T="THA4_TUNING=1"
python tools/ab_full.py --no-b1 default=default w1024=default@$T,THA4_WANT_WGS=1024 w4096=default@$T,THA4_WANT_WGS=4096 \
  w1024_m63=default@$T,THA4_WANT_WGS=1024,THA4_TILE_NW4=63 w4096_m63=default@$T,THA4_WANT_WGS=4096,THA4_TILE_NW4=63 w16k_m63=default@$T,THA4_WANT_WGS=16384,THA4_TILE_NW4=63 \
  default2=default 2>&1 | tee gpurun_out/c15_ab.txt
