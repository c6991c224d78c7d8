mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
for wl in cfg2 cfg4 cfg3 cfg5 cfg1; do
EXTRA_ARGS="--workload $wl --no-pipeline" run ${wl}_head A=1
for v in 512_48 512_0 640_48 576_48; do
EXTRA_ARGS="--workload $wl --no-pipeline" run ${wl}_occ_$v MKAMD_LIB=$R/.variants/lib_occ_$v.so
done
done
