mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base e704; do
  for wl in cfg2 cfg1 cfg4; do
    if [ $v = base ]; then L=""; else L="MKAMD_LIB=$R/.variants/libmkamd_$v.so"; fi
    env $L timeout 300 python bench.py --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $wl', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done
done
