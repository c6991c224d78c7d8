"""Print the kernel timeline (start/end in us relative to a window start) from a rocprofv3 kernel trace.

usage: python tools/timeline.py <kernel_trace.csv> [n_tile_kernels_to_skip] [n_to_show]
Used to see how the pre-pass of call n+1 overlaps the tile kernel of call n."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10
show = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tiles = [i for i, r in enumerate(rows) if "k_voxelize_tiles" in r["Kernel_Name"]]
if len(tiles) < skip + show + 1:
    skip, show = 0, len(tiles) - 1
i0, i1 = tiles[skip], tiles[skip + show]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - 12):i1 + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void mkamd::", "")[:44]
    print(f"{s:10.1f} {e:10.1f} {e - s:8.1f}  q={r.get('Queue_Id', '?'):>3} {name}")
