# round 4, call 25: c24's two changes apart - A = norm_finalize operand prefetch only, B = packed tap table only (both on top of libtha4_warm.so = warm_kernarg commit)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_full.py --rounds 2 warm=build_variants/libtha4_warm.so A_norm=build_variants/libtha4_varA.so B_taps=build_variants/libtha4_varB.so both=default 2>&1 | tee gpurun_out/c25_ab.txt
