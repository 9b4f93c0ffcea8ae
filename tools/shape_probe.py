#!/usr/bin/env python3
"""Per-shape launch times of one bench line (roofline.shapes): python tools/shape_probe.py <bench.json> [kernel prefix]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pre = sys.argv[2] if len(sys.argv) > 2 else ""
for s in d["roofline"]["shapes"]:
    if s["kernel"].startswith(pre):
        print(s["kernel"], s["M"], s["N"], s["K"], "launches", s["launches"], "avg_us %.1f" % (1e3 * s["avg_launch_ms"]), "frac %.3f" % s["frac"])
print(" ms_per_step %.1f" % d["ms_per_step"], {k.split(" ")[0]: round(1e3 * v["s"], 1) for k, v in d["roofline"]["families"].items()})
