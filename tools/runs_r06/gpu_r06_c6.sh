# round 6, call 6: in-kernel time stamps of front16r_kernel / level1_16r_kernel (where does a wave's launch go?)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_stamps.so timeout 600 python tools/stamps_student.py > gpurun_out/c6_stamps.txt 2>&1
cat gpurun_out/c6_stamps.txt
