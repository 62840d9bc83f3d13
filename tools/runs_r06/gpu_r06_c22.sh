# round 6, call 22: the round's profile capture on the final library (tools/profile_r06.sh)
set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_r06.sh > gpurun_out/profile_r06.log 2>&1
tail -5 gpurun_out/profile_r06.log
ls gpurun_out | head -80
