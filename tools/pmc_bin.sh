#!/bin/bash
# usage: tools/pmc_bin.sh <tag> <binary> [args...]  -- TCP/TA counter passes over a micro-benchmark binary (GPU box)
TAG=$1; shift; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
BIN=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
while read -r PMC; do
  [ -z "$PMC" ] && continue
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/p$i" -o p -- "$BIN" "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($PMC) failed" >> "$OUT/errors.log"
done <<'LIST'
TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM
LIST
python - "$OUT" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(root + '/summary.txt', 'w') as o:
    for k in sorted(agg):
        o.write(k + ': ' + '  '.join(f'{c}={sum(v[-3:]) / len(v[-3:]):.4g}' for c, v in sorted(agg[k].items())) + '\n')
PY
