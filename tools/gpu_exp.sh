mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
for wl in cfg3 cfg5 cfg1 cfg2; do
  for pm in -1 0; do
  timeout 300 python - $wl $pm <<'PY'
import sys, json, subprocess, os
wl, pm = sys.argv[1], sys.argv[2]
env = dict(os.environ, MKAMD_PREPASS=pm)
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--workload", wl, "--no-pipeline"], capture_output=True, text=True, env=env).stdout
d = json.loads(out.strip().splitlines()[-1]); print(wl, "prepass", pm, d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["single_grid_latency_us"])
PY
  done
done
done
