#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag> (tools/profile_round.sh) into profiles/ as round <rNN>: kernel stats, PMC summary with
the derived per-launch figures, the bench line, and the traffic entry bench.py reads.   usage: update_profiles.py <tag> <rNN>"""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, f"{rnd}_kernel_stats.csv"))
bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
json.dump(bench, open(os.path.join(dst, f"{rnd}_bench.json"), "w"), indent=1)
summ = open(os.path.join(src, "pmc_summary.txt")).read()
def val(kernel, counter):
    m = re.search(r"\] %s: [^\n]*?\b%s=([0-9.e+]+)" % (re.escape(kernel), counter), summ)
    return float(m.group(1)) if m else None
k = bench["roofline"]["kernel"]
fetch, write = val(k, "FETCH_SIZE"), val(k, "WRITE_SIZE")
waves, valu, salu = val(k, "SQ_WAVES"), val(k, "SQ_INSTS_VALU"), val(k, "SQ_INSTS_SALU")
iw, iv, isal = val("mobi_recon_intra", "SQ_WAVES"), val("mobi_recon_intra", "SQ_INSTS_VALU"), val("mobi_recon_intra", "SQ_INSTS_SALU")
rd, wr = 2 * fetch * 1024, write * 1024
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
clips = bench["config"]["clips_per_gpu"]
hdr = f"""# rocprofv3 --pmc passes (tools/profile_round.sh {tag}), per-dispatch averages over the last 8 dispatches of:
#   python bench.py --cpu-seconds 0 --no-kernel-events --e2e-clips 0 --steps 12 --warmup 4      ({clips} clips of 640x480 per launch)
# units: FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them.  Calibration on this box (tools/ubench/copy.hip, 512 MiB each way):
#   FETCH_SIZE reports 1/2 of the bytes read (all requests are 128 B), WRITE_SIZE is exact  =>  HBM read bytes = 2 * FETCH_SIZE KiB
#   cross-check: TCC_EA0_RDREQ_128B * 128 B and TCC_EA0_WRREQ_64B * 64 B give the same totals
"""
tail = f"""
# per launch of {k} ({waves:.0f} waves = octets of 8 macroblocks, default bench workload):
#   HBM read  = 2 * {fetch:.0f} KiB = {rd / 1e6:.0f} MB   HBM write = {write:.0f} KiB = {wr / 1e6:.0f} MB
#   total {(rd + wr) / 1e6:.0f} MB vs {algo / 1e6:.0f} MB algorithmic ({(rd + wr) / algo:.2f}x: 17-row x 32-byte MC windows, 1024-byte pitch with 640 used)
#   VALU {valu / waves:.0f} / SALU {salu / waves:.0f} instructions per octet
# mobi_recon_intra (one launch per step, {iw:.0f} waves = intra macroblocks): {iv / iw:.0f} VALU + {isal / iw:.0f} SALU per macroblock
"""
open(os.path.join(dst, f"{rnd}_pmc_summary.txt"), "w").write(hdr + summ + tail)
tj = os.path.join(dst, "pmc_traffic.json")
t = json.load(open(tj)) if os.path.exists(tj) else {}
t[f"B:{clips}"] = {"hbm_bytes_per_launch": int(rd + wr),
                   "source": f"profiles/{rnd}_pmc_summary.txt: 2*FETCH_SIZE + WRITE_SIZE of {k} (rocprofv3 --pmc, separate passes; FETCH_SIZE reports half the bytes on gfx950, calibrated with tools/ubench/copy.hip)"}
json.dump(t, open(tj, "w"), indent=1)
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(dst, f"{rnd}_kernel_stats.csv")))}
print("kernel stats avg us:", {n: round(float(r["AverageNs"]) / 1e3, 1) for n, r in stats.items() if "mobi" in n})
print("bench:", bench["value"], bench["ms_per_step"], bench["roofline"]["avg_launch_ms"], bench["roofline"]["frac"], "traffic", int(rd + wr))
