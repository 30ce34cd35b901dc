#!/bin/bash
# usage (GPU box): tools/exp_dparse_flags.sh  -- the device parser built with several compiler settings, parse time per 4096 clips
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_dparse.hip.o /tmp/dparse_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
run() {
  echo "== $*"
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c $P/csrc/mobi_dparse.hip -o $O/mobi_dparse.hip.o 2>&1 | grep -E "error|Spill" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/libmobiclip_hip.so || return
  timeout 200 python $REPO/tools/exp_dparse.py 4096 --device-only 2>&1 | grep -o "parse kernel.*"
}
run -O3
run -O3 -mllvm -amdgpu-sched-strategy=max-ilp
run -O3 -mllvm -amdgpu-sched-strategy=max-memory-clause
run -Os
run -O3 -fno-unroll-loops -mllvm -amdgpu-early-inline-all=true
run -O3 -DMOBI_PARSE_WAVES=5
run -O3 -mllvm -amdgpu-enable-structurizer-workarounds=false
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/dparse_keep.o $O/mobi_dparse.hip.o
