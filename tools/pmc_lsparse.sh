#!/bin/bash
# usage (GPU box): tools/pmc_lsparse.sh [clips] [K]  -- where a wave of the lock-step parser spends its life (counter passes over tools/exp_dparse.py --lockstep;
# with K: over frame-parallel groups of K frames, tools/exp_gop.py -- 24576 5 = one turn of 60 lanes per wave)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/lsp; mkdir -p "$OUT"; N=${1:-4096}; K=$2
CMD="python $REPO/tools/exp_dparse.py $N --lockstep"; [ -n "$K" ] && { CMD="python $REPO/tools/exp_gop.py $N $K 2 64"; export GOP_STEPWISE=0; }
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/p$i" -o p -- $CMD > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python $REPO/tools/pmc_summary.py "$OUT" 6 2>&1 | grep "parse_frames_ls"
