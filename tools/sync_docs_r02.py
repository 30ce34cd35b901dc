"""Rewrite the number tables of DESIGN.md (d), BASELINE.md 5 and the README paragraph from profiles/r02_*_bench.json and
profiles/r02_small_batches.jsonl, so that the prose quotes exactly what the committed profiles hold.  python tools/sync_docs_r02.py"""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = {c: json.load(open(f"{R}/profiles/r02_{c}_bench.json")) for c in "ABC"}
g = {x["config"]["clips_per_gpu"]: x for x in (json.loads(l) for l in open(f"{R}/profiles/r02_small_batches.jsonl"))}
B = d["B"]; c4 = B["config4"]; e2e = B["end_to_end"]; cpu = B["cpu_baseline"]

def row(c, name, clips):
    x = d[c]; r = x["roofline"]
    return (f"| {name} | {clips} | {x['ms_per_step']:.2f} ms | {x['value'] / 1000:.0f} | {r['avg_launch_ms']:.2f} ms | {r['frac']:.3f} | "
            f"{r['traffic'] / r['algorithmic_bytes_per_launch']:.2f} | {r['intra_kernel_ms_per_step']:.2f} ms | {r['whole_step_frac']:.3f} |")

p = f"{R}/DESIGN.md"; s = open(p).read()
for c, name, clips in (("A", "A 256×192 ModsDS", 24576), ("B", "B 640×480 Moflex3DS", 24576), ("C", "C 848×480 Moflex3DS", 6144)):
    s = re.sub(r"\| " + re.escape(name) + r" \|[^\n]*", row(c, name, clips), s, count=1)
a = s.index("B at small batches (`profiles/r02_small_batches.jsonl`)"); b = s.index("An I-frame step (all macroblocks intra,")
s = s[:a] + (f"B at small batches (`profiles/r02_small_batches.jsonl`): 64 clips {g[64]['ms_per_step']:.3f} ms per step = {g[64]['value'] / 1000:.0f} Gpixels/s "
             f"(frac {g[64]['roofline']['frac']:.2f}); 512 clips\n{g[512]['ms_per_step']:.3f} ms = {g[512]['value'] / 1000:.0f} ({g[512]['roofline']['frac']:.2f}); "
             f"4096 clips {g[4096]['ms_per_step']:.2f} ms = {g[4096]['value'] / 1000:.0f} ({g[4096]['roofline']['frac']:.2f}, whole step "
             f"{g[4096]['roofline']['whole_step_frac']:.2f}). 8 clips (`config4`): {c4['ms_per_step']:.3f} ms per step = {c4['value'] / 1000:.0f} Gpixels/s — two launches of\n"
             "1200 and ≈480 waves: launch latency plus a chain of two or three dependency levels. ") + s[b:]
open(p, "w").write(s)

def f(x): return f"{x:,.0f}".replace(",", " ")
def fr(x): r = x["roofline"]; return f"{100 * r['frac']:.1f} / {100 * r['whole_step_frac']:.1f}"
def hb(x): r = x["roofline"]; return f"{r['traffic'] / r['avg_launch_ms'] / 1e6:.0f}"
asy = e2e.get("async", {}).get("value")
e2e_txt = f"{f(e2e['value'])} ({f(asy)} asynchronous)" if asy else f(e2e["value"])
p = f"{R}/BASELINE.md"; s = open(p).read()
a = s.index("| config | GPUs | clips/GPU |")
s = s[:a] + f"""| config | GPUs | clips/GPU | Mpix/s (GPU kernel) | HBM GB/s (rocprof) | % of 8 TB/s (inter kernel / whole step) | Mpix/s (end-to-end) | CPU oracle Mpix/s (1 thr / N thr) | bit-exact |
|---|---|---|---|---|---|---|---|---|
| A 256×192 Mods P-stream | 1 | 24576 | {f(d['A']['value'])} | {hb(d['A'])} | {fr(d['A'])} | — | {cpu['value']:.0f} / — | yes |
| B 640×480 Moflex P-stream | 1 | 24576 | {f(B['value'])} | {hb(B)} | {fr(B)} | {e2e_txt} (4096 clips, device parse) | {cpu['value']:.0f} / {cpu['all_cpus']['value']:.0f} (N = 256) | yes |
| B ×8 clips (64 over 8 GPUs) | 1 | 8 | {f(c4['value'])} | — | — | — | | yes |
| B ×64 clips | 1 | 64 | {f(g[64]['value'])} | — | {fr(g[64])} | — | | yes |
| B ×512 clips | 1 | 512 | {f(g[512]['value'])} | — | {fr(g[512])} | 12 000 device / 20 500 host parse (r01 table) | | yes |
| B ×4096 clips (roofline run) | 1 | 4096 | {f(g[4096]['value'])} | — | {fr(g[4096])} | {e2e_txt} | | yes |
| C 848×480 Moflex P-stream | 1 | 6144 | {f(d['C']['value'])} | {hb(d['C'])} | {fr(d['C'])} | — | — | yes |

"""
open(p, "w").write(s)
p = f"{R}/README.md"; s = open(p).read()
a = s.index("Measured on one MI355X (round 2"); b = s.index("| read | for |")
s = s[:a] + (f"Measured on one MI355X (round 2, `python bench.py`, 640×480 Moflex3DS P-frames in stream order, 24576 resident clips, 2 s\n"
             f"timed): {B['value'] / 1000:.0f} Gpixels/s of reconstruction (command lists resident in HBM), the dominant kernel at {100 * B['roofline']['frac']:.0f} % of the 8 TB/s HBM\n"
             f"roofline counting only its own macroblocks' bytes, the whole step at {100 * B['roofline']['whole_step_frac']:.0f} %, bit-exact; {e2e['value'] / 1000:.0f} Gpixels/s end to end from\n"
             f"bitstreams in host memory with the parse on the GPU" + (f" ({asy / 1000:.0f} with two steps in flight)" if asy else "") +
             f"; {cpu['value'] / 1000:.2f} Gpixels/s for the CPU restatement of the reference on one host\n"
             f"core ({cpu['all_cpus']['value'] / 1000:.1f} on all 256). Details and the profiles behind the numbers: `DESIGN.md` § (d), `BASELINE.md` § 5, `profiles/`.\n\n") + s[b:]
open(p, "w").write(s)
print("synced")
