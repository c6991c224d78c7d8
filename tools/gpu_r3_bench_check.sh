mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_cfg2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2.log)
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; echo "rc=$?" >> gpurun_out/bench_torchrun1.log)
python - <<'PY'
import json
for n in ("cfg2", "torchrun1"):
    for l in open(f"gpurun_out/bench_{n}.log"):
        if l.startswith("{"):
            d = json.loads(l)
            print(n, {k: d.get(k) for k in ("value", "ms_per_step", "single_grid_latency_us", "gather_ms", "gather_overlapped_extra_ms", "gather_error", "sustained", "dropin_call_ms", "secondary_error")})
            print("   roofline", {k: d["roofline"][k] for k in ("frac", "kernel_avg_ms", "traffic", "step_traffic")})
            for k, v in (d.get("other_workloads") or d.get("batched_molecules") or {}).items():
                print("   ", k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error"))
    print(open(f"gpurun_out/bench_{n}.log").read()[-300:] if "rc=0" not in open(f"gpurun_out/bench_{n}.log").read() else "rc=0")
PY
