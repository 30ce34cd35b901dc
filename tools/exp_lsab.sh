#!/bin/bash
# usage (GPU box): tools/exp_lsab.sh "<flags A>" "<flags B>" ...  -- mobi_lsparse.hip built with each set of -D flags in turn, twice round, parse time of each
# (-DLS_K=n: 1..4 only -- beyond that the cheap rounds could outrun the bitstream ring and the build stops at a static_assert).  The tools load
# the profiling twin of the library (tools/_prof.py), so that is what is rebuilt; 64 distinct streams (tools/exp_dparse.py, DISTINCT).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj_prof
cp $P/libmobiclip_hip_prof.so /tmp/lib_keep.so
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o /tmp/lsv.o $O/mobi_gop.hip.o $O/mobi_analysis.hip.o"
for ROUND in 1 2; do
  for F in "$@"; do
    hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -fvisibility=hidden -DMOBI_PROFILING -mllvm -amdgpu-sched-strategy=max-ilp $F -c $P/csrc/mobi_lsparse.hip -o /tmp/lsv.o 2>&1 | grep -E " error" | head -3
    hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$O/exports.map $OBJS -o $P/libmobiclip_hip_prof.so || exit 1
    echo "[$F] $(timeout 600 python $REPO/tools/exp_dparse.py ${CLIPS:-8192} --lockstep 2>&1 | grep clips= | grep -o 'parse kernel: I [0-9.]* ms, P median [0-9.]* ms')"
  done
done
cp /tmp/lib_keep.so $P/libmobiclip_hip_prof.so
