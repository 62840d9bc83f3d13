#!/bin/bash
# round 5, call 1: stretch-barrier microbench (go/no-go of the persistent small-map kernel), bench.py line with the new fields, GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 120 tools/microbench/stretch_barrier 44 > gpurun_out/c1/stretch_barrier.txt 2>&1; echo "rc=$?" >> gpurun_out/c1/stretch_barrier.txt
timeout 60 tools/microbench/stretch_barrier 11 >> gpurun_out/c1/stretch_barrier.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; echo "bench rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/c1/pytest.txt
cat gpurun_out/c1/stretch_barrier.txt
