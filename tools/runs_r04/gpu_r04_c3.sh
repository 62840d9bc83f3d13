# round 4, call 3: four-wave workgroups (NW = 4) WITH phase priorities (s_setprio 1 in the staging phase, -DTHA4_PHASE_PRIO=1): the
# machine model says a VALU wave only hides under a partner's MFMAs at raised priority.  Also: host submit time of a full-model frame.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
T="THA4_TUNING=1"; P=build_variants/libtha4_prio.so
python tools/ab_full.py default=default prio=$P prio_all63=$P@$T,THA4_TILE_NW4=63 prio_all63_d0=$P@$T,THA4_TILE_NW4=63,THA4_TILE_DEPHASE=0 \
  prio_all63_d12k=$P@$T,THA4_TILE_NW4=63,THA4_TILE_DEPHASE=12000 prio_m32=$P@$T,THA4_TILE_NW4=32 prio_m48=$P@$T,THA4_TILE_NW4=48 \
  prio_m50=$P@$T,THA4_TILE_NW4=50 default2=default 2>&1 | tee gpurun_out/c3_ab.txt
