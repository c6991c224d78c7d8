# round 6, session 50: the feed's host side (bytes first by pread, headers out of the copy): xtc GPU tests, the leg inside the whole line
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_xtc.py -m gpu -q -x -k "xtc or XTC" 2>&1 | tail -3)
for rep in 1 2; do
  timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/bench_again.json
  python - <<'PY'
import json
c = json.load(open("gpurun_out/bench_again.json"))
x = c["other_workloads"]["xtc_cfg4"]
print(json.dumps({k: v for k, v in x["device_decode"].items()})[:1000]); print(x.get("bottleneck"), x.get("stage_ms_per_call"))
print("in line:", x["frames_per_s"], "steady", x["device_decode"].get("steady_frames_per_s"), "busy", x["gpu_busy_fraction"], "kernels alone", x["kernels_alone_frames_per_s"])
PY
done
