#!/bin/bash
# usage (GPU box): tools/exp_lsflags.sh -- mobi_lsparse.hip built with several compiler settings, parse time of each (8192 clips)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_lsparse.hip.o /tmp/l_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
run() {
  echo "== flags: $(echo $* | tr -d "-")"
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c $P/csrc/mobi_lsparse.hip -o $O/mobi_lsparse.hip.o 2>&1 | grep -E " error|spill" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/libmobiclip_hip.so || return
  timeout 600 python $REPO/tools/exp_dparse.py 8192 --lockstep 2>&1 | grep clips= | sed "s/.*parse kernel/parse kernel/"
}
run -O3
run -O3 -mllvm -amdgpu-sched-strategy=max-ilp
run -O3 -fno-unroll-loops
run -O3 -mllvm -amdgpu-sched-strategy=max-memory-clause
run -O2
run -Os
run -O3 -mllvm -amdgpu-early-ifcvt=1
run -O3 -mllvm -amdgpu-skip-uniform-regions=1
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/l_keep.o $O/mobi_lsparse.hip.o
