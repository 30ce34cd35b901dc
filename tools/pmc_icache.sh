#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r03d_pmc; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 8 --warmup 4 --cpu-seconds 0 --no-kernel-events --e2e-clips 0 --config4-clips 0 --clips 8192"
i=0
for PMC in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/p$i" -o p -- $BENCH > "$OUT/p$i.log" 2>&1 || echo "pass $i ($PMC) failed" >> "$OUT/errors.log"
done
python $REPO/tools/pmc_summary.py "$OUT" 6 > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt" | grep "inter8"; cat "$OUT/errors.log" 2>/dev/null; grep -h -i "error\|invalid\|not found" $OUT/p*.log | head -5
