#!/usr/bin/env python3
"""Prints the headline numbers of bench.py JSON lines: tools/bsum.py FILE..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {}); e = d.get("eigen_sweeps", {}); z = d.get("exact_zero_skipping", {})
        print("%s: %.1f Mpx/s %.2f ms | %s | roof %s %.0f/%.0f=%.2f avg %.3f ms x%s | sweeps %s x %.3f ms | skip %.2f ms | its %s/%s" % (
            f.split("/")[-1], d["value"], d["ms_per_step"], " ".join("%s %.1f" % (k[:4], v) for k, v in d["stage_ms_rank0"].items()),
            r.get("kernel", "?")[:18], r.get("achieved", 0), r.get("peak", 0), r.get("frac", 0), r.get("avg_launch_ms", 0), r.get("launches_per_step"),
            e.get("launches_per_step"), e.get("avg_ms", 0), z.get("ms_per_step", 0), d["config"].get("outer_its"), d["config"].get("inner_its_total")))
    except Exception as ex:
        print(f, "ERR", ex)
