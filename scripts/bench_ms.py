#!/usr/bin/env python3
"""print ms_per_step / value / launches of bench.py JSON lines given as files (helper of the scripts/gpu_*.sh A/B loops)"""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], (d.get("roofline") or {}).get("launches_per_step"))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
