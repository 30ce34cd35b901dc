#!/bin/bash
# usage: tools/pmc_valu.sh <tag>   -- VALU/SALU instruction counts of the inter kernel for several generator mixes
TAG=${1:-valu}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for GEN in "" "pm_deep=0" "pm_split1=0,pm_deep=0" "pm_split1=0,pm_deep=0,cbp_prob=0" "pm_split1=0,pm_deep=0,cbp_prob=0,pm_intra=0,mv_range=0"; do
  i=$((i+1))
  BENCH_GEN="$GEN" rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/m$i" -o p -- python $REPO/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-kernel-events > "$OUT/m$i.log" 2>&1
  echo "== mix $i: '$GEN'"
  python - "$OUT/m$i" <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if "mobi" not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in acc:
    w = acc[k]['SQ_WAVES']
    print('  ', k, 'launches', n[k], ' per wave:', {c: round(v / w, 1) for c, v in acc[k].items() if c != 'SQ_WAVES'})
PY
done
