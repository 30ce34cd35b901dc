#!/bin/bash
# usage (GPU box): tools/exp_intra_occupancy.sh -- mobi_recon_intra with fewer waves per CU (extra LDS per workgroup, profiling build): how much of
# the launch is latency that more waves would hide.  8320 B per wave = 7 granules of 1280 B = 18 waves per CU.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd
cp $P/libmobiclip_hip.so /tmp/lib_keep.so
python -m mobiclipdecoder_amd.build --profiling > /dev/null 2>&1 || { echo "profiling build failed"; exit 1; }
for PAD in 0 700 1980 3260; do
  echo "== MOBI_INTRA_LDS_PAD=$PAD ($(( 163840 / ( ( (8320 + PAD + 1279) / 1280 ) * 1280 ) )) waves per CU)"
  MOBI_INTRA_LDS_PAD=$PAD timeout 300 python $REPO/bench.py --steps 64 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 | python $REPO/tools/brief.py
done
timeout 200 python $REPO/tools/exp_iframe.py 4096 2>&1 | tail -2
cp /tmp/lib_keep.so $P/libmobiclip_hip.so
