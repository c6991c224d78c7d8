# round 6, session 62: the multi-process path on ONE GPU (ranks share the device: a rehearsal, not a scaling measurement): bench.py as the driver launches it
# for N = 2 and 4, the voxel headline and the distance row -- rendezvous, gloo fences, max over ranks, the gather legs' handling of RCCL's refusal of one device twice
mkdir -p gpurun_out
export TMPDIR=/tmp
export MKAMD_BENCH_SHARE_DEVICES=1
for n in 2 4; do
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29540 + n)) bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline --no-extra --min-seconds 1 2> gpurun_out/rehearsal_cfg2_$n.err | grep '^{' > gpurun_out/rehearsal_cfg2_$n.json; echo "cfg2 N=$n rc=${PIPESTATUS[0]}")
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29550 + n)) bench.py --workload dist --gpus $n --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/rehearsal_dist_$n.err | grep '^{' > gpurun_out/rehearsal_dist_$n.json; echo "dist N=$n rc=${PIPESTATUS[0]}")
done
python - <<'PY'
import json
for n in (2, 4):
    for wl in ("cfg2", "dist"):
        try:
            d = json.load(open(f"gpurun_out/rehearsal_{wl}_{n}.json"))
            print(wl, n, "value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "ranks_alive", d.get("ranks_alive"), "gather_error", str(d.get("gather_error"))[:80], "scaling", d.get("scaling"))
        except Exception as e:
            print(wl, n, "NO LINE:", e)
PY
tail -3 gpurun_out/rehearsal_*_4.err | cut -c1-300
