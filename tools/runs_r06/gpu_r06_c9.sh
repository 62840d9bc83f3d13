# round 6, call 9: entry / exit spans of every workgroup of the three student kernels (launch skew, tail, inter-kernel gaps)
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
THA4_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtha4_stamps.so timeout 600 python tools/stamps_student.py > gpurun_out/c9_stamps.txt 2>&1
grep -A6 "spans\|Error\|error" gpurun_out/c9_stamps.txt | head -30
