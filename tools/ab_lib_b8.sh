# same-box A/B of two builds of the library, full model batch 8 (GPU box): usage ab_lib_b8.sh <other.so> [rounds]
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/$1; N=${2:-2}
run() { python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"; }
for i in $(seq $N); do
  echo "shipped : $(run)"
  echo "other   : $(THA4_HIP_LIB=$OTHER run)"
done | tee gpurun_out/ab_lib_b8.txt
