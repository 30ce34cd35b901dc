#!/bin/bash
# usage (GPU box): VARIANTS="-DX -DY" tools/exp_variants.sh -- the reconstruction kernels rebuilt with each -D variant in turn, A/B on one box
# (r03 used it as exp_prio.sh for the s_setprio decision: -DMOBI_NO_PRIO against the kernels as they are; those macros are gone)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_kernels.hip.o /tmp/k_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
for ROUND in 1 2; do
  for F in ${VARIANTS:-"-DMOBI_NO_PRIO" "-DMOBI_AS_IS"}; do
    hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -mllvm -amdgpu-sched-strategy=max-ilp -fno-unroll-loops $F -c $P/csrc/mobi_kernels.hip -o $O/mobi_kernels.hip.o 2>&1 | grep -E " error" | head -3
    hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/libmobiclip_hip.so || exit 1
    echo "[$F] $(timeout 200 python $REPO/tools/exp_iframe.py 4096 2>&1 | tail -1 | cut -c1-60) | $(timeout 300 python $REPO/bench.py --steps 96 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 | python $REPO/tools/brief.py | cut -c1-90)"
  done
done
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/k_keep.o $O/mobi_kernels.hip.o
