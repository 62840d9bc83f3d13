# Round-5 profiles (run on the GPU box): bash tools/profile_r05.sh - the captures of rounds 2-4 (student batch 1 / batch 32, full model
# batch 1 / batch 8: rocprofv3 kernel stats, SQ PMC passes incl. the LDS bank-conflict counters, FETCH_SIZE / WRITE_SIZE passes, per-layer
# breakdown + its JSON form) on this round's FINAL library; python tools/make_profile_md.py r05 (build container) turns gpurun_out/ into profiles/r05_*.
bash tools/profile_r03.sh
