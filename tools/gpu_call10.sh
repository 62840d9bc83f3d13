set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_student_gpu.py -m gpu -x -q > gpurun_out/c10_pytest.log 2>&1; tail -3 gpurun_out/c10_pytest.log
B="python $R/bench.py --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --steps 3000 --warmup 200"
for rep in 1 2; do
for v in default nomerge r1like; do
  unset THA4_HIP_LIB
  if [ $v != default ]; then export THA4_HIP_LIB=$R/build_variants/libtha4_$v.so; fi
  $B > gpurun_out/c10_bench_$v.json 2> gpurun_out/c10_bench_$v.err; python -c "
import json; d=json.load(open('gpurun_out/c10_bench_$v.json')); print('RESULT $v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frame_event_ms'])"
done
done
unset THA4_HIP_LIB
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c10_tr -- python $R/bench.py --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --steps 1000 --warmup 100 > $R/gpurun_out/c10_tr.log 2>&1
cd $R
python tools/trace_gaps.py gpurun_out/c10_tr 600 > gpurun_out/c10_gaps.txt 2>&1; head -8 gpurun_out/c10_gaps.txt
rm -rf gpurun_out/c10_tr
