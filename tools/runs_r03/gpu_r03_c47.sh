mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 60 python -m pytest tests/test_full_gpu.py -q -x -k "all_33_outputs" 2>&1 | tail -1
