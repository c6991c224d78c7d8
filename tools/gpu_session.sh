# tools/gpu_session.sh -- the commands of the CURRENT gpurun session (rewritten from session to session; the evidence session
# that the committed profiles come from is tools/gpu_evidence.sh)
# round 5, session 4: topology reuse -- GPU tests, then cfg4 with and without the handle (same box, three rounds), and the pre-pass's counters
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
tail -6 gpurun_out/pytest_gpu.log
one() { label=$1; shift; (timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single --steps 30 --warmup 5 "$@" 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label'.ljust(34), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'], 'topo', d['config'].get('topology_reuse'))
"); }
for rep in 1 2 3; do
  one "cfg4 topology" --workload cfg4
  one "cfg4 plain" --workload cfg4 --no-topology
  one "cfg4 topology in order" --workload cfg4 --no-pipeline
  one "cfg4 plain in order" --workload cfg4 --no-topology --no-pipeline
done 2>&1 | tee gpurun_out/topology_ab.txt
PROF="--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 6 --warmup 2 --workload cfg4"
for n in topo plain; do
  extra=""; [ $n = plain ] && extra="--no-topology"
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/tpmc_$n -- python $R/bench.py $PROF $extra > $R/gpurun_out/tpmc_$n.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tstat_$n -- python $R/bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 30 --warmup 5 --workload cfg4 $extra > $R/gpurun_out/tstat_$n.log 2>&1)
done
python tools/pmc_summary.py gpurun_out/tpmc_topo gpurun_out/tpmc_plain | tee gpurun_out/topology_pmc.txt
for n in topo plain; do echo "== $n"; f=$(ls gpurun_out/tstat_$n/*/*kernel_stats.csv | head -1); head -12 $f; done | tee gpurun_out/topology_kstats.txt
