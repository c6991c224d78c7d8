mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in 0 1; do
rm -rf gpurun_out/trace_so$so
(cd /tmp && MKAMD_SPATIAL_ORDER=$so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_so$so -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline > $R/gpurun_out/trace_so$so.log 2>&1)
echo "spatial order $so"; tail -1 gpurun_out/trace_so$so.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/trace_so$so/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>1: print("  %-50s %8.1f us x %s"%(r['Name'][:50], float(r['AverageNs'])/1e3, r['Calls']))
PY
done
