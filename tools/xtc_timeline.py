#!/usr/bin/env python3
"""tools/xtc_timeline.py <dir of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/xtc_leg.py> -- per chunk of the XTC-fed leg: when its
upload, its decode (k_xtc_scan / k_xtc_expand) and its voxelization ran, and the gaps between them (what serialises the feed)."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kind = "scan" if "k_xtc_scan" in n else "expand" if "k_xtc_expand" in n else "tile" if "k_voxelize_tiles" in n else "bin" if "k_bin_count" in n else None
        if kind:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
for f in glob.glob(d + "/*/*memory_copy_trace.csv") + glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        b = int(r.get("Bytes", r.get("Size", 0)) or 0)
        if b > (8 << 20):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "h2d %d MB" % (b >> 20)))
ev.sort()
t0 = ev[0][0] if ev else 0
last = {}
for s, e, k in ev[-400:]:
    key = k.split()[0]
    gap = (s - last.get(key, s)) / 1e6
    last[key] = e
    if key in ("scan", "h2d") or (key == "tile" and gap > 0.5):
        print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e6:7.3f} ms  {k:12s} (idle before, same kind: {gap:7.3f} ms)")
