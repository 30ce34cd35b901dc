#!/bin/bash
# usage (GPU box): tools/exp_intra_occupancy.sh -- mobi_recon_intra with fewer waves per CU (extra LDS per workgroup, profiling build): how much of
# the launch is latency that more waves would hide.  8320 B per wave = 7 granules of 1280 B = 18 waves per CU.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd
export MOBI_LIB=$P/libmobiclip_hip_prof.so # the profiling twin reads MOBI_*_LDS_PAD (python -m mobiclipdecoder_amd.build --profiling; it travels with the snapshot)
for PAD in 0 700 1980 3260; do
  echo "== MOBI_INTRA_LDS_PAD=$PAD ($(( 163840 / ( ( (8320 + PAD + 1279) / 1280 ) * 1280 ) )) waves per CU)"
  MOBI_INTRA_LDS_PAD=$PAD timeout 300 python $REPO/bench.py --steps 64 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 | python $REPO/tools/brief.py
done
timeout 200 python $REPO/tools/exp_iframe.py 4096 2>&1 | tail -2
