mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
EXTRA_ARGS="--workload cfg4" run cfg4_base A=1
for v in pbc48_p3 pbc48_p0 base_p0; do
EXTRA_ARGS="--workload cfg4" run cfg4_$v MKAMD_LIB=$R/.variants/lib_$v.so
done
done
EXTRA_ARGS="--workload cfg4 --no-pipeline" run cfg4_nopipe_base A=1
EXTRA_ARGS="--workload cfg4 --no-pipeline" run cfg4_nopipe_pbc48 MKAMD_LIB=$R/.variants/lib_pbc48_p3.so
