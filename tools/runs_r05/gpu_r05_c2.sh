#!/bin/bash
# round 5, call 2: side-stream skip convolutions + XCD-aware conv_tile order: same-box A/B of the four combinations, the GPU suite on the new
# library (incl. the mid-gain fixture tests), stretch_barrier with the weight-stream modes, the root-ingest probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
timeout 200 tools/microbench/stretch_barrier 44 > gpurun_out/c2/stretch_barrier.txt 2>&1; echo "rc=$?" >> gpurun_out/c2/stretch_barrier.txt
timeout 700 python tools/ab_full.py --rounds 2 both=default noside=default@THA4_TUNING=1,THA4_NO_SIDE_STREAM=1 noremap=default@THA4_TUNING=1,THA4_NO_XCD_REMAP=1 \
   neither=default@THA4_TUNING=1,THA4_NO_SIDE_STREAM=1,THA4_NO_XCD_REMAP=1 > gpurun_out/c2/ab.txt 2>&1
cat gpurun_out/c2/ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c2/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/c2/pytest.txt
timeout 200 python tools/rccl_root_ingest_probe.py --frames 1000 > gpurun_out/c2/root_ingest.txt 2>&1; tail -4 gpurun_out/c2/root_ingest.txt
cat gpurun_out/c2/stretch_barrier.txt
