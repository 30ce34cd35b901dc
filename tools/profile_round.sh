#!/bin/bash
# Round profile (GPU box): kernel trace + the PMC passes behind the bench's roofline/traffic numbers.
# usage: tools/profile_round.sh <tag> ; outputs under gpurun_out/<tag>/ (copy the summaries into profiles/)
TAG=${1:-round}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# kernel trace: the default command (its end-to-end leg included); counter passes: the timed loop only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python $REPO/bench.py --cpu-seconds 0 > "$OUT/trace.log" 2>&1
BENCH="python $REPO/bench.py --cpu-seconds 0 --no-kernel-events --e2e-clips 0"
i=0
while read -r PMC; do
  [ -z "$PMC" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o p -- $BENCH --steps 12 --warmup 4 > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/errors.log"
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/cal$i" -o p -- $REPO/tools/ubench/copy.bin > "$OUT/cal$i.log" 2>&1 || true
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum
LIST
i=10
while read -r PMC; do
  [ -z "$PMC" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o p -- $BENCH --steps 12 --warmup 4 > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/errors.log"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr
LIST
