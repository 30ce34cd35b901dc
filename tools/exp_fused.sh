#!/bin/bash
# One launch per step (mobi_recon_step) against two, by batch size: bench.py's config-4 leg and the small-batch headline legs.
# usage (on the GPU box): tools/exp_fused.sh <tag>
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${1:-fused}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for F in 0 default; do
  if [ $F = 0 ]; then export MOBI_FUSED_STEP_MBS=0; else unset MOBI_FUSED_STEP_MBS; export MOBI_FUSED_STEP_MBS=100000000; fi
  for N in 8 64 128 256 512 1024; do
    timeout 200 python $REPO/bench.py --clips $N --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 --steps 96 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused=$F clips=$N ms_per_step', d['ms_per_step'], 'value', d['value'], 'verified', d['verified']['ok'])"
  done
done 2>&1 | tee "$OUT/fused.txt"
unset MOBI_FUSED_STEP_MBS
timeout 300 python $REPO/bench.py --clips 64 --cpu-seconds 0 --e2e-clips 0 --single-stream 1 --content-lowfreq 0 --bitmap-clips 0 --steps 96 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config4', d['config4']); print('single', d['single_stream'])" | tee -a "$OUT/fused.txt"
