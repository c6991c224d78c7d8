# round 6, session 8: the HIP-graph experiment, the RCCL collectives under torchrun with one rank (NCCL_DEBUG=WARN), the split bench.py
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 300 python tools/graph_latency.py > gpurun_out/s8_graph_latency.txt 2>&1); grep -v amdgpu.ids gpurun_out/s8_graph_latency.txt | tail -8
(NCCL_DEBUG=WARN timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --min-seconds 1 > gpurun_out/s8_bench_torchrun1.log 2> gpurun_out/s8_bench_torchrun1.err; echo "rc=$?" >> gpurun_out/s8_bench_torchrun1.log)
python - <<'PY'
import json
for l in open("gpurun_out/s8_bench_torchrun1.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step", "gather_ms", "gather_overlapped_extra_ms", "gather_exchange", "gather_error", "collectives_exercised", "ranks_alive")})
PY
tail -1 gpurun_out/s8_bench_torchrun1.log; grep -c "NCCL WARN" gpurun_out/s8_bench_torchrun1.err; grep "NCCL WARN" gpurun_out/s8_bench_torchrun1.err | head -5
(NCCL_DEBUG=WARN timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --workload dist --gpus 1 --steps 10 --warmup 3 > gpurun_out/s8_bench_dist_torchrun1.log 2> gpurun_out/s8_bench_dist_torchrun1.err; echo "rc=$?" >> gpurun_out/s8_bench_dist_torchrun1.log)
tail -2 gpurun_out/s8_bench_dist_torchrun1.log | cut -c1-1200; grep -c "NCCL WARN" gpurun_out/s8_bench_dist_torchrun1.err
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > gpurun_out/s8_bench_cfg2.log 2>&1; echo "rc=$?" >> gpurun_out/s8_bench_cfg2.log); tail -2 gpurun_out/s8_bench_cfg2.log | cut -c1-900
