mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-pipeline --steps 10 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for rep in 1 2; do
run base A=1
for v in m1b4 m1b6 m1b8 b8; do run $v MKAMD_LIB=$R/.variants/lib_$v.so; done
done
EXTRA_ARGS="--workload cfg1" run cfg1_base A=1
EXTRA_ARGS="--workload cfg1" run cfg1_m1b8 MKAMD_LIB=$R/.variants/lib_m1b8.so
