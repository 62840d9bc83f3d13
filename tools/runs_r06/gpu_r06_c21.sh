# round 6, call 21: validation of the tree (side-stream code removed, level-1 one-trip prologue, ADVICE fixes): full GPU suite, smoke, the four bench lines;
# then one planner A/B (normalisations folded into the consumer for tensors of up to 128 / 256 tiles instead of 64)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1; tail -4 gpurun_out/final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; tail -1 gpurun_out/final_smoke.log
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; cat gpurun_out/final_bench.json
timeout 900 python bench.py --model full > gpurun_out/final_bench_full.json 2> gpurun_out/final_bench_full.err; cat gpurun_out/final_bench_full.json
timeout 300 python bench.py --batch 32 --steps 60 --warmup 10 --cpu-seconds 0 > gpurun_out/final_bench_b32.json 2>/dev/null; cat gpurun_out/final_bench_b32.json
timeout 300 python bench.py --model full --batch 8 --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/final_bench_fb8.json 2>/dev/null; cat gpurun_out/final_bench_fb8.json
timeout 900 python tools/ab_full.py --rounds 2 --no-b8 default=default fuse128=default@THA4_TUNING=1,THA4_FUSED_NORM_MAX_TILES=128 fuse256=default@THA4_TUNING=1,THA4_FUSED_NORM_MAX_TILES=256 > gpurun_out/c21_ab.txt 2>&1; cat gpurun_out/c21_ab.txt
