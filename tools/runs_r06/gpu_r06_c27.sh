# round 6, call 27: what the mixed plans pay for - per-op times of the eyebrow decomposer under the three plans, batch 1 and batch 8
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/decomposer_ops.py > gpurun_out/c27_decomposer_b1.txt 2>&1; head -4 gpurun_out/c27_decomposer_b1.txt
timeout 600 python tools/decomposer_ops.py --batch 8 > gpurun_out/c27_decomposer_b8.txt 2>&1; cat gpurun_out/c27_decomposer_b8.txt
