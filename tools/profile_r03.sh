#!/bin/bash
# Round-3 profile (GPU box): for the three BASELINE configurations, the bench line, a kernel trace of the same command and the
# PMC passes behind roofline.traffic; for B also the SQ / LDS / L1 counters quoted in DESIGN.md; micro-benchmarks, small batches,
# I-frame, single stream, device parser, asynchronous steps, the GPU tests and both fuzzers.
# usage: tools/profile_r03.sh [tag]    -> gpurun_out/<tag>/{B,A,C}/...   (tools/update_profiles_r03.py copies the summaries into profiles/)
TAG=${1:-r03}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd $REPO && timeout 900 python -m pytest tests -m gpu -q > "$OUT/gpu_tests.log" 2>&1 )
for CFG in B A C; do
  D=$OUT/$CFG; mkdir -p "$D"
  EXTRA=""; [ "$CFG" != "B" ] && EXTRA="--steps 96"
  timeout -k 5 500 python $REPO/bench.py --config $CFG $EXTRA > "$D/bench.json" 2> "$D/bench.err"
  BENCH="python $REPO/bench.py --config $CFG --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0"
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o t -- $BENCH --steps 64 > "$D/trace.log" 2>&1 || echo "trace failed" >> "$D/errors.log"
  i=0
  while read -r PMC; do
    [ -z "$PMC" ] && continue
    i=$((i+1))
    if [ "$CFG" != "B" ] && [ $i -gt 2 ]; then continue; fi
    timeout -k 5 240 rocprofv3 --pmc $PMC --output-format csv -d "$D/pmc$i" -o p -- $BENCH --no-kernel-events --steps 8 --warmup 4 > "$D/pmc$i.log" 2>&1 || echo "pass $i ($PMC) failed" >> "$D/errors.log"
  done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM
TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQC_ICACHE_REQ SQC_ICACHE_MISSES
LIST
  python $REPO/tools/pmc_summary.py "$D" 6 > "$D/pmc_summary.txt" 2>&1
done
timeout 120 $REPO/tools/ubench/tilepat.bin 4096 > "$OUT/tilepat.txt" 2>&1
timeout 60 $REPO/tools/ubench/fetchpat.bin 4096 > "$OUT/fetchpat.txt" 2>&1
timeout 60 $REPO/tools/ubench/pitch.bin 640 480 16384 > "$OUT/pitch.txt" 2>&1
{ timeout 60 $REPO/tools/ubench/pwrite.bin 400000 8192; } > "$OUT/pwrite.txt" 2>&1
for N in 64 512 4096; do timeout 200 python $REPO/bench.py --clips $N --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --steps 96 2>/dev/null; done > "$OUT/bench_small.jsonl"
timeout 200 python $REPO/tools/exp_iframe.py 4096 > "$OUT/iframe.txt" 2>&1
timeout 200 python $REPO/tools/exp_dparse.py 4096 --device-only > "$OUT/dparse.txt" 2>&1
$REPO/tools/pmc_dparse.sh $TAG/dparse_pmc 4096 >> "$OUT/dparse.txt" 2>&1
# the lock-step parser (64 clips per wave) beside the one-wave-per-clip parser, at the batch sizes where each wins; its counters
{ timeout 600 python $REPO/tools/exp_dparse.py 4096 8192 24576 --device-both; $REPO/tools/pmc_lsparse.sh 4096; } > "$OUT/lsparse.txt" 2>&1
timeout 300 python $REPO/tools/exp_async.py 4096 12 > "$OUT/async.txt" 2>&1
timeout 200 python $REPO/tools/exp_rgb.py > "$OUT/rgb.txt" 2>&1
timeout 200 python $REPO/tools/exp_search.py > "$OUT/search.txt" 2>&1
{ timeout 600 python $REPO/tools/fuzz_intra_gpu.py 2000 100; timeout 600 python $REPO/tools/fuzz_inter_gpu.py 1500 100; timeout 900 python $REPO/tools/soak_parity.py 4096 B; timeout 900 python $REPO/tools/soak_parity.py 8192 B lockstep; } > "$OUT/fuzz.txt" 2>&1
ls "$OUT" > /dev/null
