# Round-4 profiles (run on the GPU box): bash tools/profile_r04.sh - the captures of rounds 2 + 3 (student batch 1 / batch 32, full model
# batch 1 / batch 8: rocprofv3 kernel stats, SQ PMC passes, FETCH_SIZE / WRITE_SIZE passes, per-layer breakdown) on this round's library;
# python tools/make_profile_md.py r04 (build container) turns gpurun_out/ into profiles/r04_*.
bash tools/profile_r03.sh
