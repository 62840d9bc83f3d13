# A/B of the v_sin_f32 sine (build_variants/libtha4_hwsin.so, -DTHA4_HW_SIN) against the shipped polynomial (GPU box)
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
X="--cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --profile-frames 0"
for lib in default build_variants/libtha4_hwsin.so; do
  if [ $lib = default ]; then unset THA4_HIP_LIB; else export THA4_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
  echo "== $lib"
  timeout 600 python -m pytest tests/test_student_gpu.py -x -q -s 2>&1 | grep -E "PARITY|passed|failed" | sort | uniq -c | sort -rn | head -12
  for r in 1 2; do python bench.py $X 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['value'])"; done
done 2>&1 | tee gpurun_out/ab_hw_sin.txt
