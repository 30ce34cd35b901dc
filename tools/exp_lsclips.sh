#!/bin/bash
# usage (GPU box): tools/exp_lsclips.sh -- the lock-step parser with 64 / 32 clips per wave (fewer lanes: fewer regions occupied per round, more waves)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_lsparse.hip.o /tmp/l_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
for N in 64 32 24; do
  echo "== $N clips per wave"
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -DLS_CLIPS=$N -c $P/csrc/mobi_lsparse.hip -o $O/mobi_lsparse.hip.o 2>&1 | grep -E " error" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/libmobiclip_hip.so || exit 1
  timeout 600 python $REPO/tools/exp_dparse.py 24576 --lockstep 2>&1 | grep clips= | cut -c150-330
done
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/l_keep.o $O/mobi_lsparse.hip.o
