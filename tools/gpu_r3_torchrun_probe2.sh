mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0 --no-gather --no-single"
run() { name=$1; shift; ("$@" > gpurun_out/tr_$name.log 2>&1); python - $name <<'PY'
import json, sys
ok = False
for l in open(f"gpurun_out/tr_{sys.argv[1]}.log"):
    if l.startswith("{"):
        d = json.loads(l); ok = True; print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f} kernel {d['roofline']['kernel_avg_ms']:.4f}")
if not ok: print(sys.argv[1], "FAILED", open(f"gpurun_out/tr_{sys.argv[1]}.log").read()[-400:])
PY
}
E="env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
run plain python bench.py $B
run nccl_eager $E MASTER_PORT=29531 MKAMD_BENCH_PG=nccl_eager python bench.py $B
run nccl_lazy $E MASTER_PORT=29532 MKAMD_BENCH_PG=nccl_lazy python bench.py $B
run mixed $E MASTER_PORT=29533 MKAMD_BENCH_PG=mixed python bench.py $B
run gloo $E MASTER_PORT=29534 MKAMD_BENCH_PG=gloo python bench.py $B
run nccl_eager_nopipe $E MASTER_PORT=29535 MKAMD_BENCH_PG=nccl_eager python bench.py $B --no-pipeline
run plain_nopipe python bench.py $B --no-pipeline
run plain2 python bench.py $B
