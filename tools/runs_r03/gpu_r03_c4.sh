set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/sin_cliff.py 0 --skip-eval --libs=hwsin_nop,hwsin_stream,hwsin_pg1 > gpurun_out/r03_sin_cliff_variants.txt 2> gpurun_out/r03_sin_cliff_variants.err; cat gpurun_out/r03_sin_cliff_variants.txt; tail -3 gpurun_out/r03_sin_cliff_variants.err
