#!/usr/bin/env python3
"""A few lines out of a bench.py JSON line: headline, roofline, every secondary workload, the CPU baseline."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(d.get("value"), d.get("unit"), "ms/step", d.get("ms_per_step"), "frac", r.get("frac"), "in flight", (r.get("all_launches_in_flight") or {}).get("frac"),
      "traffic", r.get("traffic"), "build", d.get("config", {}).get("build_id"))
for s in d.get("secondary", []):
    print(" ", s.get("name"), s.get("value"), s.get("ms_per_step"), (s.get("roofline") or {}).get("frac"), s.get("error"))
print(" cpu_baseline", d.get("cpu_baseline"))
if "summary" in d:
    print(" summary", d["summary"])
