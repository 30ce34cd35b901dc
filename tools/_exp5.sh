cd $GRAFT_REPO_ROOT
CLIPS=24576 tools/exp_lsab.sh "-DLS_CELLS_ALWAYS" "" 2>&1 | tee gpurun_out/fold4.txt
