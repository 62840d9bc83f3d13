# round 4, call 1: today's baseline + conv_tile ablations (results of abt_* / ab_mfma are wrong by construction: timing only),
# kernel stats of the shipped library at batch 1 / batch 8, and a probe of WHEN gpurun snapshots the tree
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
ls gpurun_probe_* > gpurun_out/c1_probe.txt 2>&1
V=build_variants
python tools/ab_full.py default=default abt_valu=$V/libtha4_abt_valu.so abt_window=$V/libtha4_abt_window.so abt_epi=$V/libtha4_abt_epi.so \
  abt_window_epi=$V/libtha4_abt_window_epi.so abt_all=$V/libtha4_abt_all.so ab_mfma=$V/libtha4_ab_mfma.so default2=default 2>&1 | tee gpurun_out/c1_ab.txt
cd /tmp
for cfg in "b1 --mode steady --frames 20" "b8 --batch 8 --frames 5"; do
  set -- $cfg; tag=$1; shift
  for lib in default abt_window; do
    if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$R/$V/libtha4_$lib.so; fi
    rm -rf $R/gpurun_out/c1_prof
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c1_prof -- python $R/tools/time_full.py "$@" > /dev/null 2>&1
    python $R/tools/kernel_stats.py $(ls $R/gpurun_out/c1_prof/*/*kernel_trace.csv | head -1) > $R/gpurun_out/c1_stats_${tag}_${lib}.txt
    rm -rf $R/gpurun_out/c1_prof
  done
done
unset THA4_HIP_LIB
cd $R; head -30 gpurun_out/c1_stats_b1_default.txt
