# round 4, call 16: measured cost of the pre-staged operand pass (read fp32 C16, scale/shift + SiLU + hi/lo split, write two fp16 planes)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 120 tools/microbench/prestage_pass | tee gpurun_out/c16_prestage.txt
