"""Prints the kernels of one step from a rocprofv3 --kernel-trace CSV: start (us from the step's motion kernel), duration,
idle gap before the kernel.  usage: python tools/trace_timeline.py <t_kernel_trace.csv> [step from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'motion_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = idx[-k], idx[-k + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
busy = 0
for r in rows[a:b + 1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    gap = (s - prev_end) if prev_end is not None else 0
    print("%8.1f %8.1f gap %6.1f  %s" % (s / 1e3, (e - s) / 1e3, gap / 1e3, r['Kernel_Name'][:70]))
    prev_end = e
    if r is not rows[b]: busy += e - s
print("kernel time %.1f us of %.1f" % (busy / 1e3, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
