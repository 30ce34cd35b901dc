#!/bin/bash
# usage (GPU box): tools/exp_istages.sh <tag> [clips]  -- mobi_recon_intra's dynamic instruction counts stage by stage (profiling twin,
# MOBI_INTRA_DBG = n << 8 leaves the kernel after stage n): 1 requests + dependency wait, 2 level words scattered, 3 residual transforms,
# 4 tiles zeroed + halo, 5 step list built, 6 16x16 plane + first taps asked for, 7 the steps, 0 whole kernel.  P-frame steps and,
# with IFRAME=1, an I-frame step (tools/exp_iframe.py).
TAG=${1:-istages}; CLIPS=${2:-4096}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export MOBI_LIB=$REPO/mobiclipdecoder_amd/libmobiclip_hip_prof.so
cd /tmp && export TMPDIR=/tmp
for ST in 1 2 3 4 5 6 7 0; do
  MOBI_INTRA_DBG=$((ST * 256)) timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d "$OUT/s$ST" -o p -- python $REPO/bench.py --config ${CONFIG:-B} --clips $CLIPS --steps 4 --warmup 1 --cpu-seconds 0 --no-kernel-events --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 > "$OUT/s$ST.log" 2>&1
  python - "$OUT/s$ST" $ST <<'PY'
import sys, csv, glob, collections
acc = {"P": collections.defaultdict(float), "I": collections.defaultdict(float)}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('mobi_recon_intra')]
    big = max((int(r['Grid_Size']) for r in rows), default=0)
    for r in rows:
        acc["I" if int(r['Grid_Size']) > big // 2 else "P"][r['Counter_Name']] += float(r['Counter_Value'])
for kind in "PI":
    a = acc[kind]; w = a.get('SQ_WAVES', 0) or 1
    print('stage %2s %s-frame' % (sys.argv[2], kind), ' per wave:', {c.replace('SQ_INSTS_', ''): round(v / w, 1) for c, v in sorted(a.items()) if c != 'SQ_WAVES'}, 'waves', int(w))
PY
done | tee "$OUT/summary.txt"
