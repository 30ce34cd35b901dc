cd $GRAFT_REPO_ROOT
CLIPS=4096 tools/exp_lsab.sh "-DLS_K=2" "" "-DLS_K=4" "-DLS_K=5" 2>&1 | tee gpurun_out/k4096.txt
