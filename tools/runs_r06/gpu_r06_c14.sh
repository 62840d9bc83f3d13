# round 6, call 14: mixed plans (eyebrow decomposer on the exact-fp32 kernels: whole network / outside its 16x16 bottleneck): full-model GPU tests,
# parity of the four plans on the mid-gain set, cost on steady / cold / batch-8 frames (same-box A/B through THA4_EXACT_DECOMPOSER=0|outer|all)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_full_gpu.py tests/test_api_surface.py -m gpu -q > gpurun_out/c14_pytest.log 2>&1; tail -5 gpurun_out/c14_pytest.log
for f in default split mixed_all exact; do grep "b8 up_merged\|b8 up_warped\|b8 face_6\|b8 body_merged\|b1 pose 0 up_merged" gpurun_out/full_midgain_parity_report_$f.txt | sed "s/^/$f: /"; done > gpurun_out/c14_midgain_rows.txt; cat gpurun_out/c14_midgain_rows.txt
for v in 0 outer all 0 outer all; do
  THA4_EXACT_DECOMPOSER=$v timeout 300 python bench.py --model full --cpu-seconds 0 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_and_cold']
print('exact_decomposer=$v  steady', s['steady']['fps'], ' cold', s['cold']['fps'])"
  THA4_EXACT_DECOMPOSER=$v timeout 300 python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exact_decomposer=$v  batch 8', d['value'])"
done > gpurun_out/c14_mixed_ab.txt 2>&1
grep exact_decomposer gpurun_out/c14_mixed_ab.txt
