#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py reports as roofline.traffic) from a profile's summary.json:
    python tools/update_traffic.py profiles/r04_step"""
import json
import os
import sys

d = sys.argv[1].rstrip("/")
s = json.load(open(os.path.join(d, "summary.json")))
m = s["step_kernel_main_loop"]
scale = float(1 << 20) / m["boards_per_launch"]
out = {"source": os.path.join(d, "summary.json"), "commit": s["stamp"]["git_head"], "source_digest": s["stamp"]["source_digest"],
       "boards_per_launch_measured": m["boards_per_launch"], "kernel": m["kernel"], "average_ns": m["average_ns"],
       "hbm_bytes_per_launch": m["traffic"] * scale,
       "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per step_kernel launch at 1,048,576 boards (FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md: 128-B requests are tallied at 64 B)",
       "read_requests_per_launch": m.get("read_requests"),
       "decomposed_estimate_bytes": m["traffic_decomposed"] * scale if "traffic_decomposed" in m else None,
       "decomposition": "TCC_EA0_RDREQ read requests: (32 B state + 1 B action) x boards / 128 of them are the 128-B requests of the "
                        "coalesced streams, the rest 64-B gathers (a finished board's pool record, a window refill); + WRITE_SIZE"}
old = os.path.join(os.path.dirname(d), "traffic.json")
try:
    prev = json.load(open(old))
    out["earlier_rounds_bytes_per_launch"] = dict(prev.get("earlier_rounds_bytes_per_launch", {}),
                                                  **{str(prev.get("commit")): prev.get("hbm_bytes_per_launch")})
except (OSError, ValueError):
    pass
json.dump(out, open(old, "w"), indent=1)
print(json.dumps(out, indent=1))
