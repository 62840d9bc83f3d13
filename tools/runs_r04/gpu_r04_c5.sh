# round 4, call 5: batch 8 as two / four half-batches on separate streams (host-side experiment)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 500 python tools/ab_two_streams.py 2>/dev/null | tee gpurun_out/c5_two_streams.txt
P=build_variants/libtha4_prio.so
THA4_HIP_LIB=$R/$P THA4_TUNING=1 THA4_TILE_NW4=50 timeout 500 python tools/ab_two_streams.py 2>/dev/null | tee gpurun_out/c5_two_streams_nw4prio.txt
