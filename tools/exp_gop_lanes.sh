#!/bin/bash
# Frame-parallel groups: clips per wave (L) and waves per workgroup (W) of the lock-step parser under n * K virtual clips (tools/exp_gop.py on the
# profiling twin of the library: MOBI_LS_CLIPS / MOBI_LS_WG_WAVES).   tools/exp_gop_lanes.sh [clips] [K]
cd "$(dirname "$0")/.."
export MOBI_LIB=$PWD/mobiclipdecoder_amd/libmobiclip_hip_prof.so GOP_STEPWISE=0
clips=${1:-24576}; K=${2:-6}
for lw in ${LW:-0,0 24,8 28,8 32,8 35,8 46,6}; do
  set -- ${lw/,/ }
  if [ "$1" != "0" ]; then export MOBI_LS_CLIPS=$1 MOBI_LS_WG_WAVES=$2; else unset MOBI_LS_CLIPS MOBI_LS_WG_WAVES; fi
  echo "== L=$1 W=$2"
  timeout 300 python tools/exp_gop.py $clips $K 3 64 2>&1 | grep -v "^Traceback" | tail -4
done
