# same-box A/B of two builds of the library on the full model (GPU box): usage ab_lib.sh <other.so> [rounds]
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/$1; N=${2:-3}
for i in $(seq $N); do
  echo "shipped : $(python tools/time_full.py --mode steady --frames 60 2>/dev/null | grep 'full model')"
  echo "other   : $(THA4_HIP_LIB=$OTHER python tools/time_full.py --mode steady --frames 60 2>/dev/null | grep 'full model')"
done | tee gpurun_out/ab_lib.txt
