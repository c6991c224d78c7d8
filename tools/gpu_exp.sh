mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-pipeline --steps 10 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
run base A=1
for v in nopair noflush noepi; do run $v MKAMD_LIB=$R/.variants/lib_$v.so; done
EXTRA_ARGS="--workload cfg3" run cfg3_base A=1
for v in noflush noepi; do EXTRA_ARGS="--workload cfg3" run cfg3_$v MKAMD_LIB=$R/.variants/lib_$v.so; done
