set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/compare_libs.py default build_variants/libtha4_wait0.so 1024 2>&1 | tail -1 | tee gpurun_out/c42_soak.txt
