#!/bin/bash
# round 5, call 10: small-map 1x1 convolutions of the batched plans on conv_point_kernel instead of the exact-fp32 conv_splitk_kernel fall-back: A/B at batch 8, parity of every plan
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
timeout 600 python tools/ab_full.py --rounds 3 --no-b1 point=default splitk=default@THA4_TUNING=1,THA4_NO_POINT_SMALL_MAPS=1 > gpurun_out/c10/ab.txt 2>&1
cat gpurun_out/c10/ab.txt
timeout 1200 python -m pytest tests/test_full_gpu.py tests/test_twin_gpu.py -x -q > gpurun_out/c10/pytest.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/c10/pytest.txt
