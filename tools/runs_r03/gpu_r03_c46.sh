mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 60 python tools/time_full.py > gpurun_out/pf_time.log 2>&1; tail -2 gpurun_out/pf_time.log
timeout 60 python tools/time_full.py --batch 8 --frames 20 > gpurun_out/pfb8_time.log 2>&1; tail -1 gpurun_out/pfb8_time.log
