export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_distance.py -q 2>&1 | tail -2
for lib in intree "$@"; do
L=$GRAFT_REPO_ROOT/.variants/libmkamd_$lib.so; [ $lib = intree ] && L=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
for i in 1 2; do MKAMD_LIB=$L timeout 300 python bench.py --workload dist --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib periodic', d['roofline']['kernel_avg_ms'], d['roofline']['frac'], 'nonperiodic', d['nonperiodic']['roofline']['kernel_avg_ms'], d['nonperiodic']['roofline']['frac'])
"; done; done
MKAMD_LIB=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so timeout 300 python bench.py --workload dist --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-120
