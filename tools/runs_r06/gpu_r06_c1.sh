# round 6, call 1: (a) L2 -> LDS stream microbenchmark (what bounds the weight ring), (b) the round-3 ablations rerun on the current library
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 tools/microbench/lds_stream 256 > gpurun_out/c1_lds_stream.txt 2>&1; tail -50 gpurun_out/c1_lds_stream.txt
THA4_SWEEP_VARIANTS=default,ab_fetch,ab_barrier,ab_fetch_barrier,ab_mfma,ab_zload,ab_all,ab_sin timeout 1500 python tools/sweep.py run --steps 600 > gpurun_out/c1_ablations.txt 2>&1
cat gpurun_out/c1_ablations.txt
