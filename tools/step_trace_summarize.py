"""Reduces rocprofv3's kernel + memory-copy traces to the launch sequence of the LAST complete step (a step starts at the kernel
whose name contains the given marker): name, duration, idle gap before it.   usage: step_trace_summarize.py <dir> <prefix> <marker>"""
import csv
import glob
import sys

out, prefix, marker = sys.argv[1:4]
ev = []
for f in glob.glob(f"{out}/**/{prefix}_kernel_trace.csv", recursive=True) + glob.glob(f"{out}/{prefix}_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
for f in glob.glob(f"{out}/**/{prefix}_memory_copy_trace.csv", recursive=True) + glob.glob(f"{out}/{prefix}_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev = sorted(set(ev))
starts = [i for i, e in enumerate(ev) if marker in e[2]]
if len(starts) < 3:
    print("no steps found", len(ev))
    sys.exit(0)
a, b = starts[-3], starts[-2]
t_prev = ev[a - 1][1] if a else ev[a][0]
busy = 0
for s, e, name in ev[a:b]:
    print(f"{(e - s) / 1e3:8.1f} us  gap {max(0, s - t_prev) / 1e3:6.1f}  {name}")
    busy += e - s
    t_prev = max(t_prev, e)
print(f"launches {b - a}, busy {busy / 1e3:.1f} us, span {(ev[b][0] - ev[a][0]) / 1e3:.1f} us")
