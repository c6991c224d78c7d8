mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_dropin
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_dropin -- python $R/bench.py --workload dropin --steps 30 --no-cpu-baseline > $R/gpurun_out/rocprof_dropin.log 2>&1
cd $R
ls gpurun_out/prof_dropin/*/
python - <<'PY'
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/prof_dropin/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K "+r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob("gpurun_out/prof_dropin/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"M "+r.get("Direction","")+" "+r.get("Bytes","")))
rows.sort()
t0=rows[-40][0]
for a,b,n in rows[-40:]: print(f"{(a-t0)/1e3:9.1f} {(b-t0)/1e3:9.1f} {(b-a)/1e3:7.1f}  {n}")
PY
tail -2 gpurun_out/rocprof_dropin.log | cut -c1-300
