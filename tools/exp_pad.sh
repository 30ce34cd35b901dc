#!/bin/bash
# usage (GPU box): tools/exp_pad.sh  -- mobi_recon_inter8 with N idle instructions added (stage A: vector / scalar; before the stores: vector):
# how the launch time follows the instruction count (bench line of each build, 24576 clips)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so; cp $O/mobi_kernels.hip.o /tmp/k_keep.o
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
run() {
  echo "== $*"
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -mllvm -amdgpu-sched-strategy=max-ilp -fno-unroll-loops "$@" -c $P/csrc/mobi_kernels.hip -o $O/mobi_kernels.hip.o 2>&1 | grep -E "error" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/libmobiclip_hip.so || return
  timeout 300 python $REPO/bench.py --steps 64 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 | python $REPO/tools/brief.py
}
run
run -DMOBI_EXP_PAD_A=32
run -DMOBI_EXP_PAD_A=64
run -DMOBI_EXP_PAD_A=128
run -DMOBI_EXP_PAD_D=64
run -DMOBI_EXP_PAD_S=128
cp /tmp/lib_keep.so $P/libmobiclip_hip.so; cp /tmp/k_keep.o $O/mobi_kernels.hip.o
