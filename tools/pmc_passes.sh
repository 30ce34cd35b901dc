#!/bin/bash
# usage: tools/pmc_passes.sh <tag> -- one rocprofv3 --pmc pass per line of counters below (gpurun box)
TAG=${1:-pmc}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 12 --warmup 4 --cpu-seconds 0 --no-kernel-events ${BENCH_ARGS:-}"
i=0
while read -r PMC; do
  [ -z "$PMC" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/p$i" -o p -- $BENCH > "$OUT/p$i.log" 2>&1 || echo "pass $i ($PMC) failed" >> "$OUT/errors.log"
done <<'LIST'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum
TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES
TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ
LIST
