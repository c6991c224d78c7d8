mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for wl in cfg2 cfg1; do
EXTRA_ARGS="--workload $wl" run ${wl}_lean104 A=1
EXTRA_ARGS="--workload $wl" run ${wl}_lean112 MKAMD_LIB=$R/.variants/lib_lean56.so
EXTRA_ARGS="--workload $wl" run ${wl}_lean116 MKAMD_LIB=$R/.variants/lib_lean58.so
done
EXTRA_ARGS="--workload cfg2 --no-pipeline" run cfg2_nopipe A=1
