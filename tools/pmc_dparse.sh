#!/bin/bash
# usage: tools/pmc_dparse.sh <tag> [clips]  -- instruction counts per wave (= per clip-frame) of the device-side parser
TAG=${1:-pmcdp}; CLIPS=${2:-1024}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES --output-format csv -d "$OUT/m" -o p -- python $REPO/tools/exp_dparse.py $CLIPS --device-only > "$OUT/m.log" 2>&1 </dev/null
python - "$OUT/m" <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if "mobi_parse" not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in acc:
    w = acc[k]['SQ_WAVES']
    print('  ', k, 'launches', n[k], ' per wave (all frames, I + P):', {c: round(v / w, 1) for c, v in acc[k].items() if c != 'SQ_WAVES'})
PY
