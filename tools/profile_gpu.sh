#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the bench command, CSV into gpurun_out/<tag>/.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 16 --warmup 4 --cpu-seconds 0 --no-kernel-events $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/trace.log" 2>&1
i=0
for PMC in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE32_INSTS SQ_INSTS_FLAT" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" \
  ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o p -- $BENCH > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed" >> "$OUT/errors.log"
done
ls -R "$OUT" | head -50
