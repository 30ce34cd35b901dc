#!/bin/bash
# usage (GPU box): tools/exp_stages.sh <tag> [clips]  -- dynamic instruction counts of mobi_recon_inter8 stage by stage: the profiling twin of the
# library leaves the kernel after stage n (MOBI_STOP_STAGE=n, decided at run time so nothing is optimised away); the difference of two
# runs' SQ_INSTS_* per wave is that stage's share.  Stages: 1 descriptor decode + window DMA issue, 2 level words + MV cells asked for,
# 3 waited + deep fetch issued, 4 chroma MC, 5 luma MC, 6 MC stored, 7 slow path done, 8 residual set-up, 9 tiles zeroed + scattered,
# 10 pass 1, 11 pass 2 of the first round, 12 all rounds, 0 whole kernel (with stores).
TAG=${1:-stages}; CLIPS=${2:-4096}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export MOBI_LIB=$REPO/mobiclipdecoder_amd/libmobiclip_hip_prof.so
cd /tmp && export TMPDIR=/tmp
for ST in 1 2 3 4 5 6 7 8 9 10 11 12 0; do
  MOBI_STOP_STAGE=$ST timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d "$OUT/s$ST" -o p -- python $REPO/bench.py --config ${CONFIG:-B} --clips $CLIPS --steps 4 --warmup 1 --cpu-seconds 0 --no-kernel-events --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 > "$OUT/s$ST.log" 2>&1
  python - "$OUT/s$ST" $ST <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('mobi_recon_inter8') and int(r['Grid_Size']) > 64 * 100000:  # the P-frame launches
            acc[r['Counter_Name']] += float(r['Counter_Value'])
w = acc.get('SQ_WAVES', 0) or 1
print('stage %2s' % sys.argv[2], ' per wave:', {c.replace('SQ_INSTS_', ''): round(v / w, 1) for c, v in sorted(acc.items()) if c != 'SQ_WAVES'}, 'waves', int(w))
PY
done | tee "$OUT/summary.txt"
