mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
{
for lib in default tapprefetch default tapprefetch; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_$lib.so; fi
  echo "== $lib"; timeout 40 python tools/time_full.py --frames 30 2>/dev/null | grep "full model"
done
} | tee gpurun_out/c49_tapprefetch.txt
