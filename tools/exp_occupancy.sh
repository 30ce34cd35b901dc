#!/bin/bash
# usage (GPU box): tools/exp_occupancy.sh  -- mobi_recon_inter8 with fewer waves per CU (extra LDS per workgroup, profiling build):
# how much of the launch is latency that more waves would hide (bench line of each setting, 24576 clips)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd
export MOBI_LIB=$P/libmobiclip_hip_prof.so # the profiling twin reads MOBI_*_LDS_PAD (python -m mobiclipdecoder_amd.build --profiling; it travels with the snapshot)
for PAD in 0 512 1280 3072 6144; do
  echo "== MOBI_LDS_PAD=$PAD ($(( 163840 / (10240 + PAD) )) waves per CU)"
  MOBI_LDS_PAD=$PAD timeout 300 python $REPO/bench.py --steps 64 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 | python $REPO/tools/brief.py
done
