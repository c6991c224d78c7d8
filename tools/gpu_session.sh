# round 6, session 49: chunk plans of the XTC-fed leg INSIDE the whole bench line (the place the driver reads it)
mkdir -p gpurun_out
export TMPDIR=/tmp
for plan in 2048,512; do
  MKAMD_BENCH_XTC_PLAN=$plan timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/bench_again.json
  python - <<PY
import json
c = json.load(open("gpurun_out/bench_again.json"))
x = c["other_workloads"]["xtc_cfg4"]
print(json.dumps({k: v for k, v in x["device_decode"].items()})[:900]); print(x.get("bottleneck"), x.get("stage_ms_per_call")); print("plan $plan:", x["frames_per_s"], "steady", x["device_decode"].get("steady_frames_per_s"), "busy", x["gpu_busy_fraction"], "kernels alone", x["kernels_alone_frames_per_s"])
PY
done | tee gpurun_out/xtc_plans_in_line.txt
