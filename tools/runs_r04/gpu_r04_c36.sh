# round 4, call 36: bench.py exactly as the contract's default (no flags): one JSON line, wall clock
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
T0=$(date +%s); python bench.py > gpurun_out/c36_bench_noflags.json 2> gpurun_out/c36_bench_noflags.err; T1=$(date +%s); echo "wall $((T1-T0)) s"; wc -l gpurun_out/c36_bench_noflags.json; head -c 500 gpurun_out/c36_bench_noflags.json; echo
