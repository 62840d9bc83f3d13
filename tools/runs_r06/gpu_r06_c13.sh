# round 6, call 13: (a) boundary microbenchmark with the L2 warm-up forms 9 / 10; (b) mixed plan (eyebrow decomposer on the exact-fp32 kernels):
# parity of the three plans on the mid-gain set, cost on steady / cold / batch-8 frames (same box A/B through THA4_EXACT_DECOMPOSER)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 tools/microbench/stretch_barrier 44 > gpurun_out/c13_stretch_barrier.txt 2>&1; cat gpurun_out/c13_stretch_barrier.txt
timeout 1200 python -m pytest tests/test_full_gpu.py -m gpu -x -q -k "midgain or exact_fp32_plan or decomposer_cache or every_launch_plan" > gpurun_out/c13_pytest.log 2>&1; tail -3 gpurun_out/c13_pytest.log
for f in mixed split exact; do grep "b8 up_merged\|b8 up_warped\|b8 face_6\|b1 pose 0 up_merged" gpurun_out/full_midgain_parity_report_$f.txt | sed "s/^/$f: /"; done
for v in 0 1 0 1; do
  THA4_EXACT_DECOMPOSER=$v timeout 300 python bench.py --model full --cpu-seconds 0 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_and_cold']
print('exact_decomposer=$v  steady', s['steady']['fps'], ' cold', s['cold']['fps'])"
  THA4_EXACT_DECOMPOSER=$v timeout 300 python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exact_decomposer=$v  batch 8', d['value'])"
done 2>&1 | tee gpurun_out/c13_mixed_ab.txt
