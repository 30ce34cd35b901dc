#!/bin/bash
# usage (GPU box): VARIANTS="-DX=0 -DX=1,-DY=2" [ROUNDS=2] [CLIPS=24576] tools/exp_ab.sh -- mobi_kernels.hip rebuilt with each -D variant in turn, A/B on ONE box:
# I-frame step at 4096 and CLIPS clips (tools/exp_iframe.py) and the headline P-frame line (bench.py, no side legs)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_kernels.hip.o /tmp/k_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
for ROUND in $(seq 1 ${ROUNDS:-2}); do
  for V in $VARIANTS; do
    F=$(echo $V | tr ',' ' ')  # (a variant may be several -D flags joined by commas)
    hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -O3 -mllvm -amdgpu-sched-strategy=max-ilp -fno-unroll-loops $F -c $P/csrc/mobi_kernels.hip -o $O/mobi_kernels.hip.o 2>&1 | grep -E " error" | head -3
    hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$O/exports.map $OBJS -o $P/libmobiclip_hip.so || exit 1
    echo "[$V] $(timeout 200 python $REPO/tools/exp_iframe.py 4096 2>&1 | tail -1 | cut -c9-75) | $(timeout 200 python $REPO/tools/exp_iframe.py ${CLIPS:-24576} 2>&1 | tail -1 | cut -c9-75) | $(timeout 300 python $REPO/bench.py --clips ${CLIPS:-24576} --steps 96 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --bitmap-clips 0 --content-lowfreq 0 | python $REPO/tools/brief.py | cut -c1-100)"
  done
done
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/k_keep.o $O/mobi_kernels.hip.o
