mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
EXTRA_ARGS="--no-pipeline" run cfg2_general_new MKAMD_FORCE_GENERAL=1
EXTRA_ARGS="--no-pipeline" run cfg2_general_head MKAMD_FORCE_GENERAL=1 MKAMD_LIB=$R/.variants/lib_head.so
EXTRA_ARGS="--no-pipeline" run cfg2_general_old MKAMD_FORCE_GENERAL=1 MKAMD_LIB=$R/.variants/lib_old.so
EXTRA_ARGS="--no-pipeline" run cfg2_new A=1
EXTRA_ARGS="--no-pipeline" run cfg2_head MKAMD_LIB=$R/.variants/lib_head.so
