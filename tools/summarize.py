"""Print a compact summary of gpurun_out/ (bench JSON lines + rocprof kernel stats)."""
import csv, glob, json, sys
for f in sorted(glob.glob('gpurun_out/bench_*.log')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l); r = d.get('roofline', {'frac': None})
            print(f"{f.split('/')[-1]:28s} value={d['value']:>10} ms/step={d['ms_per_step']:<7} tile_ms={r.get('kernel_avg_ms')!s:<8} frac={r.get('frac')!s:<8} single_us={d.get('single_grid_latency_us')}")
for f in sorted(glob.glob('gpurun_out/prof_*/*/*_kernel_stats.csv')):
    print(f)
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:58]:58s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={r['Percentage']}")
