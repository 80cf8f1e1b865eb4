#!/usr/bin/env python3
"""Least-squares model of the sector-query kernel's vector instructions per particle:
wave-instructions = a + b * (obstacle cells within z_max) + c * beams, from the four counter runs of
tools/floor_fit.sh.     python tools/floor_fit.py gpurun_out/<tag> <issue-cycles per instruction> <out.json>"""
import csv, glob, json, os, re, sys
import numpy as np
src, cpi, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rows = []
for log in sorted(glob.glob(os.path.join(src, "fit_*.log"))):
    m = re.search(r"cells_in_range ([\d.]+) occupancy ([\d.]+) beams (\d+)", open(log).read())
    d = log[:-4]
    vals = {}          # per instantiation (the query may run as two launches per step: their instructions add up)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "raycast_kernel<float, false, false, false" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
                vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    rows.append({"cells_in_range": float(m.group(1)), "occupancy": float(m.group(2)), "beams": int(m.group(3)),
                 "valu_wave_instructions_per_particle": sum(float(np.mean(v)) for v in vals.values()) / (1 << 20),
                 "launches": {k[:90]: len(v) for k, v in vals.items()}})
A = np.array([[1.0, r["cells_in_range"], r["beams"]] for r in rows])
y = np.array([r["valu_wave_instructions_per_particle"] for r in rows])
coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
fit = A @ coef
res = {"runs": rows, "model": "wave-instructions per particle = a + b * cells_in_range + c * beams",
       "a": coef[0], "b_per_cell": coef[1], "c_per_beam": coef[2], "fit": list(fit), "worst_relative_residual": float(np.max(np.abs(fit - y) / y)),
       "issue_cycles_per_instruction": cpi,
       "floor_ms_per_2pow20_particles": {"%d cells x %d beams" % (r["cells_in_range"], r["beams"]):
                                         f * cpi * (1 << 20) / 1024 / 2.4e9 * 1e3 for r, f in zip(rows, fit)}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
