# round 4, call 18: does THA4_PHASE_PRIO (on by default since this round, for the full model's four-wave tiles) cost the student anything?
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 400 --warmup 100 --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --batched-steps 0 --repeats 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['repeats']['all'], {k: round(v*1000,1) for k,v in j['roofline']['kernel_ms'].items() if v})"; }
for i in 1 2 3; do
  echo "prio on  (shipped): $(run)"
  echo "prio off          : $(THA4_HIP_LIB=$R/build_variants/libtha4_noprio.so run)"
done | tee gpurun_out/c18_student_prio.txt
