#!/bin/bash
# usage (GPU box): tools/exp_lsab_gop.sh "<flags A>" "<flags B>" ...  -- as tools/exp_lsab.sh (mobi_lsparse.hip rebuilt into the profiling twin with each set of
# -D flags, twice round), measured under frame-parallel groups: CLIPS x K virtual clips (default 24576 x 5 = one turn of 60 lanes per wave), the parse
# kernels by events and the pipelined ms per frame step (tools/exp_gop.py on the twin: its reconstruction kernels carry stage-stop tests).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj_prof
cp $P/libmobiclip_hip_prof.so /tmp/lib_keep.so
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o /tmp/lsv.o $O/mobi_gop.hip.o $O/mobi_analysis.hip.o"
export MOBI_LIB=$P/libmobiclip_hip_prof.so GOP_STEPWISE=0
for ROUND in 1 2; do
  for F in "$@"; do
    hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -fvisibility=hidden -DMOBI_PROFILING -mllvm -amdgpu-sched-strategy=max-ilp $F -c $P/csrc/mobi_lsparse.hip -o /tmp/lsv.o 2>&1 | grep -E " error" | head -3
    hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$O/exports.map $OBJS -o $P/libmobiclip_hip_prof.so || exit 1
    echo "[$F] $(timeout 600 python $REPO/tools/exp_gop.py ${CLIPS:-24576} ${K:-5} 4 64 2>&1 | grep -E "parse kernels|pipelined" | sed -e 's/.*parse kernels (events) \([0-9.]*\) ms/parse \1 ms;/' -e 's/.*: \([0-9.]* ms per frame step\).*/pipelined \1/' | tr '\n' ' ')"
  done
done
cp /tmp/lib_keep.so $P/libmobiclip_hip_prof.so
