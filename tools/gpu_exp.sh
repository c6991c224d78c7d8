mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for wl in cfg2 cfg4; do
EXTRA_ARGS="--workload $wl" run ${wl}_paced A=1
EXTRA_ARGS="--workload $wl" run ${wl}_prio3 MKAMD_LIB=$R/.variants/lib_head.so
EXTRA_ARGS="--workload $wl" run ${wl}_prio0 MKAMD_LIB=$R/.variants/lib_prio0.so
done
done
EXTRA_ARGS="--workload cfg2 --batch 32" run cfg2_b32_paced A=1
EXTRA_ARGS="--workload cfg2 --batch 32" run cfg2_b32_prio3 MKAMD_LIB=$R/.variants/lib_head.so
EXTRA_ARGS="--workload cfg2 --batch 64" run cfg2_b64_paced A=1
EXTRA_ARGS="--workload cfg2 --batch 64" run cfg2_b64_prio3 MKAMD_LIB=$R/.variants/lib_head.so
EXTRA_ARGS="--workload cfg2 --no-pipeline" run cfg2_nopipe A=1
