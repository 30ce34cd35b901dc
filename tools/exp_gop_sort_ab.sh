#!/bin/bash
# Kernel totals of pipelined frame-parallel groups with the intra macroblocks launched in wavefront order (mobi_launch_gop_sort) against the
# raster-order launch (mobi_recon_intra_cl): MOBI_GOP_INTRA_SORT=1 / 0 on the profiling twin, under rocprofv3 --kernel-trace.
#   gpurun --timeout 900 -- 'bash tools/exp_gop_sort_ab.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for s in 1 0; do
  rm -rf /tmp/prof$s
  MOBI_LIB=$R/mobiclipdecoder_amd/libmobiclip_hip_prof.so MOBI_GOP_INTRA_SORT=$s GOP_STEPWISE=0 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$s -o t -- python $R/tools/exp_gop.py 24576 5 6 64 2>&1 | grep -E "Gpixels"
  f=$(find /tmp/prof$s -name "*kernel_stats.csv" | head -1)
  echo "== sort=$s ($f)"
  [ -n "$f" ] && head -14 "$f" | cut -d, -f1-6
done
