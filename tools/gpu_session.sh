# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 17: the clock probe (mkamd_clock_probe_dev): its test, and the default bench line with `sustained.shader_clock_ghz_under_load`
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "clock_probe" 2>&1 | tail -3)
(timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/bench_probe.log 2>&1); python - <<'PY'
import json
for l in open("gpurun_out/bench_probe.log"):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["sustained"])
PY
