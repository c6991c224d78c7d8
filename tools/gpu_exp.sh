mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in base m1; do
rm -rf $R/gpurun_out/pmcx_$v
if [ $v = m1 ]; then export MKAMD_LIB=$R/.variants/lib_m1.so; fi
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcx_$v -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/pmcx_$v.log 2>&1
done
unset MKAMD_LIB
cd $R
python tools/pmc_summary.py gpurun_out/pmcx_base gpurun_out/pmcx_m1 | grep -i "tiles<8" | cut -c1-400
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-pipeline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for rep in 1 2; do run base A=1; run m1 MKAMD_LIB=$R/.variants/lib_m1.so; done
