# round 3, GPU call 2: sin cliff with hand-off dumps, new hand-off test, s_setprio phase-priority A/B (student + full)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/sin_cliff.py 0 > gpurun_out/r03_sin_cliff.txt 2> gpurun_out/r03_sin_cliff.err; tail -16 gpurun_out/r03_sin_cliff.txt; tail -3 gpurun_out/r03_sin_cliff.err
timeout 600 python -m pytest tests/test_student_gpu.py -m gpu -q -s -k "hand_off or fused_display" 2>&1 | grep -E "passed|failed|PARITY hand|Error" | tail
X="--cpu-seconds 0 --d2h-frames 0 --exact-frames 0"
for lib in default prio; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_$lib.so; fi
  for r in 1 2; do timeout 300 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib student fps', d['value'], d['roofline']['kernel_ms'], 'full', d['full_model']['steady']['fps'], d['full_model']['cold']['fps'])"; done
  timeout 300 python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib full b8 fps', d['value'])"
done 2>&1 | tee gpurun_out/c2_prio_ab.txt
unset THA4_HIP_LIB
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_prio.so timeout 900 python -m pytest tests/test_student_gpu.py tests/test_full_gpu.py -m gpu -q -x -k "output0_parity or all_33_outputs_vs_reference or batch_equals" 2>&1 | tail -3
