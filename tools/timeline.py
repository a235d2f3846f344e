#!/usr/bin/env python3
"""Print the merged host-API / kernel / memcpy timeline of the last evaluations of a rocprofv3 run
(rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py ...)."""
import csv, glob, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api ", r["Function"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERN", r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r.get("Direction", "") + " " + r.get("Size", "")))
ev.sort()
anchor = [e for e in ev if e[2] == "KERN" and (len(sys.argv) <= 3 or sys.argv[3] in e[3])]
if len(sys.argv) > 3 and anchor:          # window around the last launch of the named kernel
    a = anchor[-1]
    ev = [e for e in ev if a[0] - 400_000 <= e[0] <= a[1] + 80_000]
else:
    ev = ev[-n_last:]
t0 = ev[0][0]
for s, e, k, name in ev:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {k} {name}")
