#!/bin/bash
# usage (GPU box): tools/exp_kernel_flags.sh  -- mobi_kernels.hip built with several compiler settings, bench line of each (8192 clips)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_kernels.hip.o /tmp/k_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
run() {
  echo "== $*"
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c $P/csrc/mobi_kernels.hip -o $O/mobi_kernels.hip.o 2>&1 | grep -E "error" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/libmobiclip_hip.so || return
  timeout 300 python $REPO/bench.py --clips 8192 --steps 96 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 | python $REPO/tools/brief.py
}
BASE="-O3 -mllvm -amdgpu-sched-strategy=max-ilp -fno-unroll-loops"
run $BASE
run $BASE -fno-slp-vectorize
run $BASE -mllvm -amdgpu-early-ifcvt=1
run $BASE -mllvm -amdgpu-skip-uniform-regions=1
run $BASE -mllvm -amdgpu-late-structurize=1
run $BASE -fno-vectorize -fno-slp-vectorize
run $BASE -mllvm -amdgpu-scalar-ir-passes=0
run $BASE -mllvm -amdgpu-enable-pre-ra-optimizations=0
run -O3 -fno-unroll-loops
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/k_keep.o $O/mobi_kernels.hip.o
