#!/bin/bash
# usage: [CLIPS=4096] [BENCH_ARGS=...] [GROUPS="1 2 5"] tools/pmc_quick.sh <tag>
# One rocprofv3 --pmc pass per counter group below on a small batch (GPU box).  Every pass runs under `timeout`: an invalid
# counter combination makes rocprofv3 abort and then hang in its signal handler.
TAG=${1:-pq}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 8 --warmup 4 --cpu-seconds 0 --no-kernel-events --e2e-clips 0 --clips ${CLIPS:-4096} ${BENCH_ARGS:-}"
i=0
while read -r PMC; do
  [ -z "$PMC" ] && continue
  i=$((i+1))
  if [ -n "$GROUPS_SEL" ] && ! echo " $GROUPS_SEL " | grep -q " $i "; then continue; fi
  timeout -k 5 ${PASS_TIMEOUT:-120} rocprofv3 --pmc $PMC --output-format csv -d "$OUT/p$i" -o p -- $BENCH > "$OUT/p$i.log" 2>&1 || echo "pass $i ($PMC) failed" >> "$OUT/errors.log"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM
TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
LIST
python $REPO/tools/pmc_summary.py "$OUT" 6 > "$OUT/summary.txt" 2>&1
