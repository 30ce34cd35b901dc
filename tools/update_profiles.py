#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag> (tools/profile.sh) into profiles/ as round <rNN>.  For each BASELINE configuration:
the bench line, rocprofv3's kernel stats, the same trace broken down by kernel AND grid size (the full-size launches of the timed
region apart from the I-frame re-seeds and the small legs), the PMC summary with per-launch HBM traffic; plus the micro-benchmark
outputs DESIGN.md quotes.   usage: update_profiles.py <tag> <rNN>"""
import collections, csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, RND = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
tj = os.path.join(dst, "pmc_traffic.json")
traffic = {}
sys.path.insert(0, ROOT)
import bench
KSHA = bench.kernels_sha16()  # bench.py quotes the traffic only while the kernel sources it was measured with are unchanged
for cfg in "BAC":
    d = os.path.join(src, cfg)
    bench = json.loads([l for l in open(os.path.join(d, "bench.json")) if l.startswith("{")][-1])
    json.dump(bench, open(os.path.join(dst, f"{RND}_{cfg}_bench.json"), "w"), indent=1)
    shutil.copy(os.path.join(d, "trace", "t_kernel_stats.csv"), os.path.join(dst, f"{RND}_{cfg}_kernel_stats.csv"))
    # per kernel and grid size
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, "trace", "t_kernel_trace.csv"))):
        if "mobi" in r["Kernel_Name"]:
            groups[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(os.path.join(dst, f"{RND}_{cfg}_kernel_by_grid.txt"), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace of: python bench.py --config {cfg} --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --content-lowfreq 0 --bitmap-clips 0 --steps 64\n")
        o.write("# (I-frame at start + 8 warm-up + 64 timed P-frame steps + 2 I-frame re-seeds; grid = work-items = 64 x waves)\n")
        o.write("# kernel                grid_size    launches   avg_ms    min_ms    max_ms\n")
        for (k, g), v in sorted(groups.items()):
            o.write(f"{k:22s} {g:12d} {len(v):8d} {sum(v) / len(v):9.4f} {min(v):9.4f} {max(v):9.4f}\n")
    summ = open(os.path.join(d, "pmc_summary.txt")).read()
    def val(kernel, counter):
        m = re.search(r"\] %s: [^\n]*?\b%s=([0-9.e+]+)" % (re.escape(kernel), counter), summ)
        return float(m.group(1)) if m else None
    clips = bench["config"]["clips_per_gpu"]
    lines = [f"# rocprofv3 --pmc passes (tools/profile.sh), per-dispatch averages over the last 6 dispatches of:",
             f"#   python bench.py --config {cfg} --cpu-seconds 0 --e2e-clips 0 --config4-clips 0 --single-stream 0 --no-kernel-events --steps 8 --warmup 4   ({clips} clips per launch)",
             "# one counter group per run; FETCH_SIZE / WRITE_SIZE in KiB.  gfx950: FETCH_SIZE reports half of the bytes read (128-byte requests",
             "# tallied at 64 B: MI355X_MICROARCH.md, re-checked with tools/ubench/copy.hip in r01), WRITE_SIZE is exact => HBM bytes = 2*FETCH + WRITE", ""]
    tail = []
    for k in ("mobi_recon_inter8", "mobi_recon_intra"):
        f, w = val(k, "FETCH_SIZE"), val(k, "WRITE_SIZE")
        if f is None or w is None:
            continue
        hbm = (2 * f + w) * 1024
        tail.append(f"# {k}: HBM read {2 * f * 1024 / 1e6:.0f} MB + write {w * 1024 / 1e6:.0f} MB = {hbm / 1e6:.0f} MB per launch")
        if k == "mobi_recon_inter8":
            algo = bench["roofline"]["algorithmic_bytes_per_launch"]
            tail[-1] += f" = {hbm / algo:.2f} x the {algo / 1e6:.0f} MB of algorithmic bytes"
            traffic[f"{cfg}:{clips}"] = {"hbm_bytes_per_launch": int(hbm), "kernels_sha16": KSHA, "source": f"profiles/{RND}_{cfg}_pmc_summary.txt: 2*FETCH_SIZE + WRITE_SIZE of mobi_recon_inter8 "
                                         "(rocprofv3 --pmc, separate passes; FETCH_SIZE reports half the bytes on gfx950)"}
        if k == "mobi_recon_intra" and f"{cfg}:{clips}" in traffic:
            traffic[f"{cfg}:{clips}"]["intra_hbm_bytes_per_launch"] = int(hbm)  # (bench.py: roofline.intra.traffic)
        wv, va, sa = val(k, "SQ_WAVES"), val(k, "SQ_INSTS_VALU"), val(k, "SQ_INSTS_SALU")
        if wv:
            tail.append(f"#   {wv:.0f} waves; per wave: {va / wv:.0f} VALU + {sa / wv:.0f} SALU instructions"
                        + (f", {val(k, 'SQ_INSTS_VMEM') / wv:.1f} VMEM, {val(k, 'SQ_INSTS_LDS') / wv:.0f} LDS" if val(k, "SQ_INSTS_VMEM") else "")
                        + (f", {val(k, 'TCP_TCC_READ_REQ_sum') / wv:.0f} read + {val(k, 'TCP_TCC_WRITE_REQ_sum') / wv:.0f} write requests L1->L2" if val(k, "TCP_TCC_READ_REQ_sum") else ""))
    open(os.path.join(dst, f"{RND}_{cfg}_pmc_summary.txt"), "w").write("\n".join(lines) + summ + "\n" + "\n".join(tail) + "\n")
    r = bench["roofline"]
    print(cfg, "value", bench["value"], "ms/step", bench["ms_per_step"], "inter ms", r["avg_launch_ms"], "frac", r["frac"], "whole", r["whole_step_frac"], "traffic", traffic.get(f"{cfg}:{clips}", {}).get("hbm_bytes_per_launch"))
json.dump(traffic, open(tj, "w"), indent=1)
with open(os.path.join(dst, f"{RND}_ubench.txt"), "w") as o:
    for name in ("stages_inter", "stages_intra", "intra_ablate", "iframe", "fused", "hostparse", "lsparse", "async", "gop", "gop_lanes", "gop_sort_ab", "allhost", "soak_gop", "fuzz"):
        p = os.path.join(src, name + ".txt")
        if os.path.exists(p):
            o.write(f"==== {name} ====\n" + open(p).read() + "\n")
if os.path.exists(os.path.join(src, "bench_small.jsonl")):
    shutil.copy(os.path.join(src, "bench_small.jsonl"), os.path.join(dst, f"{RND}_small_batches.jsonl"))
if os.path.exists(os.path.join(src, "gpu_tests.log")):
    shutil.copy(os.path.join(src, "gpu_tests.log"), os.path.join(dst, f"{RND}_gpu_tests.log"))
