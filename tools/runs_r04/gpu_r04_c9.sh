# round 4, call 9: in-kernel stamps of four-wave conv_tile workgroups at batch 8 (how two co-resident workgroups actually interleave),
# the exact-fp32 plan's speed, and the forced-wait twin tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/time_full_exact.py 2>/dev/null | tee gpurun_out/c9_exact_plan.txt
timeout 400 python tools/phase_timing_full.py --batch 8 --targets "tile=256x256 cin=128(cb 8) cout=128;tile=128x128 cin=256(cb 16) cout=256;tile=512x512 cin=64(cb 4) cout=64" 2>/dev/null | tee gpurun_out/c9_phase_b8.txt
THA4_TUNING=1 THA4_TILE_NW4=0 timeout 400 python tools/phase_timing_full.py --batch 8 --targets "tile=256x256 cin=128(cb 8) cout=128" 2>/dev/null | tee gpurun_out/c9_phase_b8_nw8.txt
timeout 900 python -m pytest tests/test_twin_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/c9_twin.txt
