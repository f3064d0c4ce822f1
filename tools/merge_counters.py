#!/usr/bin/env python3
"""Merges the workloads of gpurun_out/counters.json (written by tools/gpu_counters.sh) into profiles/counters.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
new = json.load(open(os.path.join(ROOT, "gpurun_out", "counters.json")))
dst = os.path.join(ROOT, "profiles", "counters.json")
old = json.load(open(dst))
keys = sys.argv[1:] or list(new["workloads"])
for k in keys:
    old["workloads"][k] = new["workloads"][k]
json.dump(old, open(dst, "w"), indent=1, sort_keys=True)
print("merged", keys)
