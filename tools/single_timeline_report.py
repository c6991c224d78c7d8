#!/usr/bin/env python3
"""tools/single_timeline_report.py DIR -- per-kernel start offsets / durations of the last calls in a rocprofv3
kernel trace of tools/single_timeline.py (median over the last 10 calls)."""
import csv, glob, os, sys, statistics
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a call starts at the memset / first mkamd kernel after a gap > 15 us
calls, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 15000 and cur:
        calls.append(cur); cur = []
    cur.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mkamd::", "")[:44], s, e))
    last_end = e
if cur: calls.append(cur)
calls = [c for c in calls if any("voxelize" in k[0] for k in c)][-10:]
names = [k[0] for k in calls[-1]]
print(f"{'kernel':46s} {'start':>8s} {'dur':>8s} {'gap before':>10s}   (us, median of {len(calls)} calls)")
for i, nm in enumerate(names):
    st = [c[i][1] - c[0][1] for c in calls if len(c) == len(names)]
    du = [c[i][2] - c[i][1] for c in calls if len(c) == len(names)]
    gp = [c[i][1] - c[i - 1][2] if i else 0 for c in calls if len(c) == len(names)]
    print(f"{nm:46s} {statistics.median(st)/1e3:8.1f} {statistics.median(du)/1e3:8.1f} {statistics.median(gp)/1e3:10.1f}")
tot = [c[-1][2] - c[0][1] for c in calls]
print(f"first kernel start -> last kernel end: {statistics.median(tot)/1e3:.1f} us")
