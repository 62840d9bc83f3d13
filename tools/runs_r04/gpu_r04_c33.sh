# round 4, call 33: folded-norm table with 16 tile moments in flight (conv_tile, conv_point, conv_small<1>), phase-2 launches skip the staging set-up: A/B against c29's library
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 3 --no-b8 prev=build_variants/libtha4_c29.so new=default 2>&1 | tee gpurun_out/c33_ab.txt
