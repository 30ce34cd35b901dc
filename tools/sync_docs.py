#!/usr/bin/env python3
"""Rewrites the number-carrying parts of DESIGN.md (d), BASELINE.md section 5 and the README paragraph from profiles/r03_* (after
tools/update_profiles_r03.py).  Text around the numbers lives here, so that a re-profile of a later build keeps the documents honest."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
B, A, C = (json.load(open(f"profiles/r03_{c}_bench.json")) for c in "BAC")
T = json.load(open("profiles/pmc_traffic.json"))
small = [json.loads(l) for l in open("profiles/r03_small_batches.jsonl")]
summ = open("profiles/r03_B_pmc_summary.txt").read()
m = re.search(r"# mobi_recon_inter8: HBM read (\d+) MB \+ write (\d+) MB = (\d+) MB per launch = ([0-9.]+) x", summ)
rd_gb, wr_gb, ratio = int(m.group(1)) / 1e3, int(m.group(2)) / 1e3, float(m.group(4))
m = re.search(r"per wave: (\d+) VALU \+ (\d+) SALU instructions, ([0-9.]+) VMEM, (\d+) LDS, (\d+) read \+ (\d+) write", summ)
valu, salu, vmem, lds, rreq, wreq = m.groups()
mi = re.findall(r"per wave: (\d+) VALU \+ (\d+) SALU", summ)[1]
grid = [l.split() for l in open("profiles/r03_B_kernel_by_grid.txt") if l.startswith("mobi_recon_inter8")]
trace_ms = max((int(g[2]), float(g[3])) for g in grid)[1]
iframe = re.search(r"I-frame, 4096 clips: ([0-9.]+) ms", open("profiles/r03_ubench.txt").read())
iframe_ms = float(iframe.group(1)) if iframe else float("nan")

def row(name, d, clips, key, r02):
    r = d["roofline"]
    return (f"| {name} | {clips} | {d['config']['stream_ms_per_step']:.2f} ms | {d['value'] / 1e3:.0f} | {r['avg_launch_ms']:.2f} ms | {r['frac']:.3f} | "
            f"{T[key]['hbm_bytes_per_launch'] / r['algorithmic_bytes_per_launch']:.2f} | {r['intra_kernel_ms_per_step']:.2f} ms | {r['whole_step_frac']:.3f} | {r02} |")

ss, e2e, c4, cb = B["single_stream"], B["end_to_end"], B["config4"], B["cpu_baseline"]
e2l = B.get("end_to_end_large") or {"value": 0, "ms_per_step": 0, "clips": 0, "async": {"value": 0, "ms_per_step": 0}}
s = open("DESIGN.md").read()
a, b = s.index("Results, MI355X, r03 build"), s.index("Profiles: `profiles/r03_*`")
text = f'''Results, MI355X, r03 build (`profiles/r03_{{A,B,C}}_bench.json`; the rocprofv3 kernel-trace average of the same command,
`profiles/r03_B_kernel_by_grid.txt`: {trace_ms:.2f} ms for B's full-size launches against {B['roofline']['avg_launch_ms']:.2f} in the untraced line):

| config | clips | step | Gpixels/s | `mobi_recon_inter8` | roofline frac | HBM traffic ÷ algorithmic | `mobi_recon_intra` | whole-step frac | r02: step / frac / traffic / whole |
|---|---|---|---|---|---|---|---|---|---|
{row("A 256×192 ModsDS", A, 24576, "A:24576", "1.63 ms / 0.364 / 1.32 / 0.306")}
{row("B 640×480 Moflex3DS", B, 24576, "B:24576", "9.74 ms / 0.379 / 1.36 / 0.319")}
{row("C 848×480 Moflex3DS", C, 6144, "C:6144", "4.23 ms / 0.307 / 1.35 / 0.274")}

Box-to-box spread of the same build is ±1 % with an occasional slow box (before the wave priorities, five boxes: 8.48, 8.48, 8.49, 8.53 and once 8.83 ms per step of B; with them, four boxes: 8.26–8.43). What the
counters say about B (`profiles/r03_B_pmc_summary.txt`): per octet {valu} VALU + {salu} SALU instructions, {float(vmem):.0f} vector-memory and {lds} LDS
instructions, {rreq} read + {wreq} write requests L1→L2 (r02: 125 + 48), HBM read {rd_gb:.1f} GB + write {wr_gb:.1f} GB per launch = {ratio:.2f} × the
{B['roofline']['algorithmic_bytes_per_launch'] / 1e9:.1f} GB of algorithmic bytes (r02: 1.36 ×; the rest is 128-byte lines of windows no neighbour shares),
`SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES` ≈ 1: the launch is bound by vector issue. `mobi_recon_intra`: {mi[0]} VALU + {mi[1]} SALU per wave of four macroblocks.

B at small batches (`profiles/r03_small_batches.jsonl`): ''' + "; ".join(
    f"{x['config']['clips_per_gpu']} clips {x['ms_per_step']:.3f} ms per step = {x['value'] / 1e3:.0f} Gpixels/s (inter frac {x['roofline']['frac']:.2f}, whole step {x['roofline']['whole_step_frac']:.2f})"
    for x in small) + f'''. 8 clips
(`config4`: BASELINE.json's 64 clips over 8 GPUs = 8 per GPU): {c4['ms_per_step']:.3f} ms per step = {c4['value'] / 1e3:.0f} Gpixels/s — two launches of 1200 and ≈480
waves: launch latency plus a chain of two or three dependency levels. An I-frame step (all macroblocks intra, outside the timed region):
{iframe_ms:.1f} ms at 4096 clips (r02: 6.1).

**Single stream** (`single_stream`, VERDICT r02 "missing" 5): what the boundary replaces is one `MobiclipDecoder` used by one thread
(`MobiConverter/Program.cs:57-71`, `Form1.cs:199-215`). One 640×480 clip through `mobi_create` / `mobi_decode` per frame (host parse,
upload, two launches, synchronise), wall time per call: **P-frame {ss['planes']['p_frame_ms']:.2f} ms, I-frame {ss['planes']['i_frame_ms']:.2f} ms**; with `mobi_get_argb` (the Bitmap `DecodeFrame()`
returns) {ss['with_bitmap']['p_frame_ms']:.2f} / {ss['with_bitmap']['i_frame_ms']:.2f} ms; the oracle on one host thread: {ss['oracle_ms_per_frame_1_thread']:.2f} ms per frame (planes only). One clip fills 0.3 % of the part: a
P-frame is two launches of 150 + ≈60 waves, an I-frame a chain of 1200 dependent macroblocks — the GPU is 3 × faster than one host core on
P-frames and 2 × slower on I-frames; its place is the batch.

`cpu_baseline`: the oracle (a C restatement, expected to be faster than the C# original: no GC, no per-row allocations), same
stream, parse + reconstruction, one C call per clip: one thread {cb['value']:.0f} Mpixels/s on the GPU box's host (`value`); one thread including
the Bitmap conversion {cb['with_bitmap']['value']:.0f} Mpixels/s (`with_bitmap`); `all_cpus` = one decoder per host cpu ({cb['all_cpus']['cores']}): {cb['all_cpus']['value'] / 1e3:.1f} Gpixels/s.

End to end (`mobi_batch_decode`: bitstream bytes in host memory → planes in HBM; staging, H2D, device parse, reconstruction,
read-back of 32 B per clip, synchronisation): {e2e['ms_per_step']:.1f} ms per step of 4096 clips = {e2e['value'] / 1e3:.0f} Gpixels/s (`end_to_end`), 10 × below the
reconstruction kernels; `mobi_batch_submit` / `mobi_batch_wait` with two steps in flight: {e2e['async']['ms_per_step']:.1f} ms = {e2e['async']['value'] / 1e3:.0f} Gpixels/s (`end_to_end.async`).
At the headline batch with the lock-step parser in front (`end_to_end_large`, {e2l['clips']} clips, f3): {e2l['ms_per_step']:.1f} ms per step = **{e2l['value'] / 1e3:.0f} Gpixels/s**,
asynchronous {e2l['async']['ms_per_step']:.1f} ms = **{e2l['async']['value'] / 1e3:.0f} Gpixels/s** (r02: a batch of that size could not be parsed on the GPU at all). Of
the 4096-clip step the parse kernel is 10.9 ms (one wave per clip, unchanged in r03: § "Next rows", f3), reconstruction 1.5 ms. PCIe-inclusive rate of the
*reconstruction* path fed with host-parsed command lists: ≈90 KB of commands per 640×480 frame, 10 % of the pixel bytes, 20 Gpixels/s with
32 parse threads (the parse, not PCIe, limits).

'''
s = s[:a] + text + s[b:]
open("DESIGN.md", "w").write(s)

def gb(d, key):
    return T[key]["hbm_bytes_per_launch"] / d["roofline"]["avg_launch_ms"] / 1e6
def n(v):
    return format(v, ",.0f").replace(",", " ")
rows = [f"| A 256×192 Mods P-stream | 1 | 24576 | {n(A['value'])} | {gb(A, 'A:24576'):.0f} | {A['roofline']['frac'] * 100:.1f} / {A['roofline']['whole_step_frac'] * 100:.1f} | — | {A['cpu_baseline']['value']:.0f} / — | yes |",
        f"| B 640×480 Moflex P-stream | 1 | 24576 | {n(B['value'])} | {gb(B, 'B:24576'):.0f} | {B['roofline']['frac'] * 100:.1f} / {B['roofline']['whole_step_frac'] * 100:.1f} | {n(e2e['value'])} ({n(e2e['async']['value'])} asynchronous) (4096 clips, device parse); {n(e2l['value'])} ({n(e2l['async']['value'])}) at {e2l['clips']} clips, lock-step parse | {cb['value']:.0f} / {cb['all_cpus']['value']:.0f} (N = {cb['all_cpus']['cores']}) | yes |",
        f"| C 848×480 Moflex P-stream | 1 | 6144 | {n(C['value'])} | {gb(C, 'C:6144'):.0f} | {C['roofline']['frac'] * 100:.1f} / {C['roofline']['whole_step_frac'] * 100:.1f} | — | {C['cpu_baseline']['value']:.0f} / — | yes |",
        f"| B ×8 clips (64 over 8 GPUs) | 1 | 8 | {n(c4['value'])} | — | — | — | | yes |"]
for x in small:
    k = x["config"]["clips_per_gpu"]
    rows.append(f"| B ×{k} clips | 1 | {k} | {n(x['value'])} | — | {x['roofline']['frac'] * 100:.1f} / {x['roofline']['whole_step_frac'] * 100:.1f} | "
                + (f"{n(e2e['value'])} ({n(e2e['async']['value'])} asynchronous)" if k == 4096 else "—") + " | | yes |")
rows.append(f"| B single stream (`mobi_decode`, one clip) | 1 | 1 | {n(ss['value'])} (P-frame {ss['planes']['p_frame_ms']:.2f} ms, I-frame {ss['planes']['i_frame_ms']:.2f} ms per call; with the Bitmap "
            f"{ss['with_bitmap']['p_frame_ms']:.2f} / {ss['with_bitmap']['i_frame_ms']:.2f} ms) | — | — | = | {307.2 / ss['oracle_ms_per_frame_1_thread']:.0f} / — ({ss['oracle_ms_per_frame_1_thread']:.2f} ms per frame) | yes |")
s = open("BASELINE.md").read()
a = s.index("| config | GPUs | clips/GPU |")
s = s[:a] + "| config | GPUs | clips/GPU | Mpix/s (GPU kernel) | HBM GB/s (rocprof) | % of 8 TB/s (inter kernel / whole step) | Mpix/s (end-to-end) | CPU oracle Mpix/s (1 thr / N thr) | bit-exact |\n|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n"
open("BASELINE.md", "w").write(s)
s = open("README.md").read()
a, b = s.index("Measured on one MI355X (round 3"), s.index("| read | for |")
s = s[:a] + f'''Measured on one MI355X (round 3, `python bench.py`, 640×480 Moflex3DS P-frames in stream order, 24576 resident clips, 1.7 s
timed): {B['value'] / 1e3:.0f} Gpixels/s of reconstruction (command lists resident in HBM), the dominant kernel at {B['roofline']['frac'] * 100:.0f} % of the 8 TB/s HBM
roofline counting only its own macroblocks' bytes, the whole step at {B['roofline']['whole_step_frac'] * 100:.0f} %, HBM traffic {ratio:.2f} × the algorithmic bytes, bit-exact;
{e2e['value'] / 1e3:.0f} Gpixels/s end to end from bitstreams in host memory with the parse on the GPU ({e2e['async']['value'] / 1e3:.0f} with two steps in flight) at 4096
clips, {e2l['value'] / 1e3:.0f} ({e2l['async']['value'] / 1e3:.0f}) at {e2l['clips']} clips with the lock-step parser; one
stream through `mobi_decode`: {ss['planes']['p_frame_ms']:.2f} ms per P-frame; {cb['value'] / 1e3:.2f} Gpixels/s for the CPU restatement of the reference on one host
core ({cb['all_cpus']['value'] / 1e3:.1f} on all {cb['all_cpus']['cores']}). Since round 3 the planes live in HBM as macroblock tiles (`mobi_tile.h`): the reference's linear
offsets keep their meaning through a bit permutation, and a macroblock's samples are three whole 128-byte lines.
Details and the profiles behind the numbers: `DESIGN.md` § (d), `BASELINE.md` § 5, `profiles/`.

''' + s[b:]
open("README.md", "w").write(s)
print("synced: B", B["value"], B["ms_per_step"], B["roofline"]["frac"], B["roofline"]["whole_step_frac"])
