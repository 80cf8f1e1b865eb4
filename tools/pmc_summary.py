#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 counter passes collected by tools/profile_round.sh.

    python tools/pmc_summary.py gpurun_out/prof_r01 profiles/r01_pmc_per_kernel.json
"""
import csv, glob, json, os, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {}
for k, ctrs in acc.items():
    out[k] = {c: {"launches": len(d), "per_launch_mean": sum(d.values()) / len(d)} for c, d in sorted(ctrs.items())}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in out:
    if "raycast" in k:
        print(k)
        for c, v in out[k].items():
            print("   %-22s %16.1f  (%d launches)" % (c, v["per_launch_mean"], v["launches"]))
