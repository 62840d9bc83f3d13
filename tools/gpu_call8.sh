set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
B="python $R/bench.py --cpu-seconds 0 --full-frames 0 --d2h-frames 0 --exact-frames 0 --steps 1000 --warmup 100"
for v in fold nofold; do
  if [ $v = nofold ]; then export THA4_HIP_LIB=$R/build_variants/libtha4_nofold.so; fi
  $B > gpurun_out/c8_bench_$v.json 2> gpurun_out/c8_bench_$v.err; python -c "
import json; d=json.load(open('gpurun_out/c8_bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frame_event_ms'])"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c8_tr_$v -- $B > $R/gpurun_out/c8_tr_$v.log 2>&1
  cd $R
  python tools/trace_gaps.py gpurun_out/c8_tr_$v 600 > gpurun_out/c8_gaps_$v.txt 2>&1; head -9 gpurun_out/c8_gaps_$v.txt
  rm -rf gpurun_out/c8_tr_$v
done
