#!/bin/bash
# round 5, call 6: moment accumulators (producers' atomics + consumer-side folding instead of 54 norm_finalize launches per batch-1 frame): same-box A/B, GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
timeout 700 python tools/ab_full.py --rounds 3 acc=default finalize=default@THA4_TUNING=1,THA4_NO_MOMENT_ACC=1 > gpurun_out/c6/ab.txt 2>&1
cat gpurun_out/c6/ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c6/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/c6/pytest.txt
