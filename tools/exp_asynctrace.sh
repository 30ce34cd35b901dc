#!/bin/bash
# usage (GPU box): tools/exp_asynctrace.sh [clips]  -- asynchronous steps with the lock-step parser in front under rocprofv3 --kernel-trace: when do the
# parse of step n + 1 and the reconstruction of step n run, and what does each cost the other?  (r05: they start together and take 35 and 17 + 4 ms
# instead of 26 and 7.5 + 1.7: the step is no shorter than parse + reconstruction one after the other.)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-24576}; OUT=$REPO/gpurun_out/asynctrace; mkdir -p $OUT
export LOCKSTEP=1
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/exp_async.py $N 6 > $OUT/trace.log 2>&1
grep -E "asynchronous|synchronous" $OUT/trace.log
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
big = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 300000]
print("kernels longer than 0.3 ms: name start end duration (ms)")
for r in big:
    print("  %-28s %9.2f %9.2f  %7.2f" % (r["Kernel_Name"][:28], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
