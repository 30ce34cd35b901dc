#!/bin/bash
# usage (GPU box): tools/exp_occupancy.sh  -- mobi_recon_inter8 with fewer waves per CU (extra LDS per workgroup, profiling build):
# how much of the launch is latency that more waves would hide (bench line of each setting, 24576 clips)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd
cp $P/libmobiclip_hip.so /tmp/lib_keep.so
python -m mobiclipdecoder_amd.build --profiling > /dev/null 2>&1 || { echo "profiling build failed"; exit 1; }
for PAD in 0 512 1280 3072 6144; do
  echo "== MOBI_LDS_PAD=$PAD ($(( 163840 / (10240 + PAD) )) waves per CU)"
  MOBI_LDS_PAD=$PAD timeout 300 python $REPO/bench.py --steps 64 --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 | python $REPO/tools/brief.py
done
cp /tmp/lib_keep.so $P/libmobiclip_hip.so
