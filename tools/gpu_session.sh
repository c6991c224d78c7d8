# round 6, session 11: cdist / pdist row kernels (0.26 / 0.15 of the roofline before), GPU distance tests
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -3)
(timeout 600 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/s11_bench_dist.log 2>&1; echo "rc=$?" >> gpurun_out/s11_bench_dist.log)
python - <<'PY'
import json
for l in open("gpurun_out/s11_bench_dist.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("cdist_pdist", json.dumps(d["cdist_pdist"]))
PY
tail -1 gpurun_out/s11_bench_dist.log
