# why is the tile kernel ~6 % slower under torchrun with one rank?  same box, back to back
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0"
run() { name=$1; shift; ("$@" > gpurun_out/tr_$name.log 2>&1); python - $name <<'PY'
import json, sys
for l in open(f"gpurun_out/tr_{sys.argv[1]}.log"):
    if l.startswith("{"):
        d = json.loads(l); print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f} kernel {d['roofline']['kernel_avg_ms']:.4f} single {d.get('single_grid_latency_us')}")
PY
}
run plain python bench.py $B
run plain_omp1 env OMP_NUM_THREADS=1 python bench.py $B
run rank_env env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 python bench.py $B
run rank_env_nogather env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29522 python bench.py $B --no-gather
run torchrun python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29523 bench.py --gpus 1 $B
run plain2 python bench.py $B
