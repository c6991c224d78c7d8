# round 6, session 34: rows per group from a rounds model (25 rows: two full rounds): tests, A-B, per-kernel times
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_distance.py -m gpu -q -x -k "contact or pair_table" 2>&1 | tail -3)
for rep in 1 2; do
  MKAMD_ALLOW_DIAGNOSTICS=1 MKAMD_LIB=$GRAFT_REPO_ROOT/.variants/libmkamd_prev.so timeout 300 python tools/pair_walk_ab.py 2>&1 | grep -v amdgpu
  timeout 300 python tools/pair_walk_ab.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/pair_walk_ab.txt
for only in "all wrap" "none wraps" "30 chains"; do
  rm -rf gpurun_out/prof_pw
  (cd /tmp && PAIR_WALK_ONLY="$only" timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pw -o pw --output-format csv -- python $GRAFT_REPO_ROOT/tools/pair_walk_ab.py > /dev/null 2>&1)
  echo "== $only"
  python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_pw/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "contacts" in r["Name"] or "k_dist_pairs" in r["Name"] or "k_build" in r["Name"]:
            print("   ", r["Name"].split("(")[0][-50:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
rm -rf gpurun_out/prof_pw
