# round 4, call 27: the first weight request of conv_small_kernel / conv_tile_kernel in front of the per-lane set-up: same-box A/B against libtha4_epi.so (= c26's library)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 prev=build_variants/libtha4_epi.so new=default 2>&1 | tee gpurun_out/c27_ab.txt
