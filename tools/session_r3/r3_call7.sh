mkdir -p gpurun_out/r3
cd /tmp
bash $GRAFT_REPO_ROOT/tools/step_timeline.sh r3s1 --stage 1 > /dev/null 2>&1
bash $GRAFT_REPO_ROOT/tools/step_timeline.sh r3s2 --stage 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/tl_r3s1.txt gpurun_out/tl_r3s2.txt
