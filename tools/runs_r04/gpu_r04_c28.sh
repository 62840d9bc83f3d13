# round 4, call 28: one hipGraph per frame of the full model vs stream launches (probe)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/graph_probe_full.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c28_graph.txt
