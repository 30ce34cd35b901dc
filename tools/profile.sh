#!/bin/bash
# The round's profile (GPU box): for the three BASELINE configurations the bench line (timed region >= 1 s each), a kernel trace of the same
# command and the PMC passes behind roofline.traffic; for B also the SQ / LDS / L1 counters DESIGN.md quotes and the stage-by-stage
# instruction counts of both reconstruction kernels; small batches, I-frame, device parsers, asynchronous steps, fuzzers and a soak.
# usage: tools/profile.sh <tag>    -> gpurun_out/<tag>/{B,A,C}/...   then (in the container) tools/update_profiles.py <tag> <rNN>
TAG=${1:-prof}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd $REPO && timeout 900 python -m pytest tests -m gpu -q > "$OUT/gpu_tests.log" 2>&1 )
for CFG in B A C; do
  D=$OUT/$CFG; mkdir -p "$D"
  # A: 1.3 ms per step, C: 12 ms (24576 clips, halved by bench.py if they do not fit): >= 1 s of timed region for each
  EXTRA=""; [ "$CFG" = "A" ] && EXTRA="--steps 1024"; [ "$CFG" = "C" ] && EXTRA="--steps 400"
  timeout -k 5 900 python $REPO/bench.py --config $CFG $EXTRA > "$D/bench.json" 2> "$D/bench.err"
  BENCH="python $REPO/bench.py --config $CFG --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0"
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o t -- $BENCH --steps 64 > "$D/trace.log" 2>&1 || echo "trace failed" >> "$D/errors.log"
  i=0
  while read -r PMC; do
    [ -z "$PMC" ] && continue
    i=$((i+1))
    if [ "$CFG" != "B" ] && [ $i -gt 2 ]; then continue; fi
    timeout -k 5 300 rocprofv3 --pmc $PMC --output-format csv -d "$D/pmc$i" -o p -- $BENCH --no-kernel-events --steps 8 --warmup 4 > "$D/pmc$i.log" 2>&1 || echo "pass $i ($PMC) failed" >> "$D/errors.log"
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
bash $REPO/tools/exp_stages.sh $TAG/stages_inter 4096 > "$OUT/stages_inter.txt" 2>&1
bash $REPO/tools/exp_istages.sh $TAG/stages_intra 4096 > "$OUT/stages_intra.txt" 2>&1
timeout 400 python $REPO/tools/exp_intra_ablate.py 8192 > "$OUT/intra_ablate.txt" 2>&1
for N in 64 512 4096; do timeout 200 python $REPO/bench.py --clips $N --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 --steps 96 2>/dev/null; done > "$OUT/bench_small.jsonl"
timeout 200 python $REPO/tools/exp_iframe.py 4096 > "$OUT/iframe.txt" 2>&1
timeout 200 python $REPO/tools/exp_iframe.py 24576 >> "$OUT/iframe.txt" 2>&1
bash $REPO/tools/exp_fused.sh $TAG/fused_raw > "$OUT/fused.txt" 2>&1
{ for N in 256 512 1024 2048; do timeout 200 python $REPO/tools/exp_hostparse.py $N | tail -1; done; MOBI_HOST_CHUNK=0 timeout 200 python $REPO/tools/exp_hostparse.py 1024 | tail -1 | sed "s/^/no pipeline: /"; MOBI_PARSE_THREADS=32 timeout 200 python $REPO/tools/exp_hostparse.py 1024 | tail -1; } > "$OUT/hostparse.txt" 2>&1
{ timeout 600 python $REPO/tools/exp_dparse.py 4096 8192 24576 --device-both; } > "$OUT/lsparse.txt" 2>&1
timeout 300 python $REPO/tools/exp_async.py 4096 12 > "$OUT/async.txt" 2>&1
{ for N in 8192 24576 49152; do echo "== $N clips, lock-step parser in front"; LOCKSTEP=1 timeout 300 python $REPO/tools/exp_async.py $N 8 | grep -E "^asynchronous|^synchronous"; done; } >> "$OUT/async.txt" 2>&1
# r06: frame-parallel groups (mobi_batch_decode_gop / gop_begin + gop_finish) end to end, the lock-step parser's lanes x waves under n * K virtual clips,
# and the two all-host situations (a ModsDS batch below quantiser 12; every clip handed over in one frame of an asynchronous batch)
# (the product library: the profiling twin's kernels carry their stage-stop tests; one run on the twin for where a group's time goes)
{ for NK in "256 6 128 107" "512 6 128 107" "1024 6 128 107" "1024 6 32 24" "1024 6 12 9" "2048 6 64 64" "2048 6 32 24" "4096 6 32 24" "4096 6 12 9" "4096 6 6 9" "8192 6 15 12" "8192 6 6 9" "16384 4 8 9" "24576 5 5 9" "24576 4 4 9" "24576 6 6 9"; do set -- $NK; GOP_KP=$3 timeout 400 python $REPO/tools/exp_gop.py $1 $2 $4 64; done; # (clips, K of the synchronous call, frames per gop_begin, groups of K: at least three timed groups of the pipelined part)
  echo "== on the profiling twin (events around the parse kernels)"; MOBI_LIB=$REPO/mobiclipdecoder_amd/libmobiclip_hip_prof.so GOP_STEPWISE=0 timeout 400 python $REPO/tools/exp_gop.py 24576 5 5 64;
  MOBI_LIB=$REPO/mobiclipdecoder_amd/libmobiclip_hip_prof.so GOP_STEPWISE=0 timeout 400 python $REPO/tools/exp_gop.py 4096 6 5 64; } > "$OUT/gop.txt" 2>&1
( cd $REPO && LW="24,8 28,8 35,8 64,4 15,8 24,4" timeout 900 tools/exp_gop_lanes.sh 24576 6 ) > "$OUT/gop_lanes.txt" 2>&1
timeout 700 bash $REPO/tools/exp_gop_sort_ab.sh > "$OUT/gop_sort_ab.txt" 2>&1
timeout 600 python $REPO/tools/exp_allhost.py > "$OUT/allhost.txt" 2>&1
{ timeout 900 python $REPO/tools/soak_parity.py 4096 B gop | tail -8; } > "$OUT/soak_gop.txt" 2>&1
{ timeout 600 python $REPO/tools/fuzz_intra_gpu.py 2000 100; timeout 600 python $REPO/tools/fuzz_inter_gpu.py 3000 100; timeout 900 python $REPO/tools/soak_parity.py 4096 B; timeout 900 python $REPO/tools/soak_parity.py 8192 B lockstep; timeout 900 python $REPO/tools/exp_refusals.py 1500 --gpu; } > "$OUT/fuzz.txt" 2>&1
ls "$OUT" > /dev/null
