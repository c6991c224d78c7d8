# round 6, session 37: the new choice among the three rectangular kernels: GPU distance tests, the shape probe (free choice), the dist line
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -3)
PROBE_AVOID=0 PROBE_SHAPES=300x30,300x60,30x300,60x300,300x100,100x300,300x150,150x300,300x200,100x100,64x640,1000x30,1000x60,2000x100,5000x60,300x300,200x500 timeout 600 python tools/dist_shapes_probe.py 2>&1 | grep -v amdgpu | cut -c1-170 | tee gpurun_out/dist_choice.txt
(timeout 600 python bench.py --workload dist --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/dist_line.json)
python - <<'PY'
import json
d = json.load(open("gpurun_out/dist_line.json"))
print("periodic rows", d["ms_per_step"], d["roofline"]["frac"], "open", d["nonperiodic"]["ms_per_step"], d["nonperiodic"]["roofline"]["frac"])
for k in ("nonperiodic", "periodic"):
    print("selfdist", k, d["selfdist"][k]["us_per_call"], d["selfdist"][k]["frac"], "small", d["small_call"][k]["us_per_call"], d["small_call"][k]["kernel"])
print("contacts", d["contacts"]["ms_per_call"], d["contacts"]["pair_tests_per_s_G"])
PY
