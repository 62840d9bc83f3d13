# round 4, call 12: planner knobs at batch 1 - four-wave tiles also on single-round grids (>= 512 workgroups), folded normalisation up to 128 tiles
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
T="THA4_TUNING=1"
python tools/ab_full.py --rounds 3 --no-b8 default=default min512=default@$T,THA4_TILE_NW4_MIN_WGS=512 fuse128=default@$T,THA4_FUSED_NORM_MAX_TILES=128 \
  both=default@$T,THA4_TILE_NW4_MIN_WGS=512,THA4_FUSED_NORM_MAX_TILES=128 2>&1 | tee gpurun_out/c12_ab.txt
