#!/bin/bash
# usage (GPU box): tools/gop_trace.sh [clips] [K] [out]  -- what a frame-parallel group's launches cost: rocprofv3 --kernel-trace over tools/exp_gop.py
# (the product library), every kernel by grid size -> gpurun_out/gop_kernel_by_grid.txt (profiles/rNN_gop_kernel_by_grid.txt is a copy)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-24576}; K=${2:-5}; OUT=${3:-$REPO/gpurun_out/gop_kernel_by_grid.txt}
cd /tmp && export TMPDIR=/tmp GOP_STEPWISE=0
rm -rf /tmp/goptrace
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/goptrace -o t -- python $REPO/tools/exp_gop.py $N $K 6 64 > /tmp/goptrace.log 2>&1
python - "$N" "$K" "$OUT" <<'PY'
import collections, csv, glob, sys
n, k, out = sys.argv[1:4]
groups = collections.defaultdict(list)
for f in glob.glob("/tmp/goptrace/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mobi" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0]
            g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1)
            groups[(name, 0 if name in ("mobi_recon_intra", "mobi_recon_intra_cl", "mobi_gop_sort_fill") else g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(out, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats -- python tools/exp_gop.py {n} {k} 6 64   (GOP_STEPWISE=0): frame-parallel groups of {k} x {n} clips of 640x480, the I-frame on its own,\n")
    o.write("# seven synchronous groups, then pipelined ones.  mobi_recon_intra and mobi_gop_sort_fill have one grid size per step (the step's intra macroblocks): summed up as grid 0.\n")
    o.write("# kernel                   grid_size    launches   avg_ms    min_ms    max_ms\n")
    for (name, g), v in sorted(groups.items()):
        o.write(f"{name:24s} {g:12d} {len(v):8d} {sum(v) / len(v):9.4f} {min(v):9.4f} {max(v):9.4f}\n")
print(open(out).read())
PY
grep -E "pipelined|decode_gop" /tmp/goptrace.log | cut -c1-160
