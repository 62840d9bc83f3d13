# round 4, call 4: where a conv_tile_kernel workgroup's cycles go today (in-kernel stamps incl. the epilogue) and whether the batch-1
# stream is host-bound (submit time of short bursts into an empty queue)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/host_submit_full.py 2>/dev/null | tee gpurun_out/c4_host_submit.txt
timeout 400 python tools/phase_timing_full.py 2>/dev/null | tee gpurun_out/c4_phase.txt
