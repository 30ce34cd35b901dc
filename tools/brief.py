#!/usr/bin/env python3
"""stdin: output of bench.py -> one short line (step ms, dominant-kernel ms, roofline frac, value)."""
import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        r = d.get("roofline") or {}
        print("ms_per_step", d["ms_per_step"], "kernel_ms", r.get("avg_launch_ms"), "frac", r.get("frac"), "value", d["value"],
              "intra_ms", r.get("intra_kernel_ms_per_step"), "e2e", (d.get("end_to_end") or {}).get("value"),
              "e2e_large", (d.get("end_to_end_large") or {}).get("value"), "async", ((d.get("end_to_end_large") or {}).get("async") or {}).get("value"))
