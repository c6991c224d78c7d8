# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 6, session 3: the reduction kernel with the next step's loads issued before this step's arithmetic (register pairs of the same step)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -3)
for i in 1 2; do
(MKAMD_DIST_ONLY=reduction timeout 600 python bench.py --workload dist --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/s3_bench_red_$i.log 2>&1; echo "rc=$?" >> gpurun_out/s3_bench_red_$i.log)
tail -2 gpurun_out/s3_bench_red_$i.log | cut -c1-1500
done
(timeout 600 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/s3_bench_dist.log 2>&1; echo "rc=$?" >> gpurun_out/s3_bench_dist.log)
python - <<'PY'
import json
for l in open("gpurun_out/s3_bench_dist.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("reduction", json.dumps(d["reduction"])); print("contacts", json.dumps(d["contacts"]))
PY
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH"
rm -rf gpurun_out/pmc_red_*
(cd /tmp && MKAMD_DIST_ONLY=reduction timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/pmc_red_sq1 -- python $R/bench.py --workload dist --no-cpu-baseline --settle-seconds 0 --steps 4 --warmup 1 > $R/gpurun_out/pmc_red_sq1.log 2>&1)
(cd /tmp && MKAMD_DIST_ONLY=reduction timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $R/gpurun_out/pmc_red_sq2 -- python $R/bench.py --workload dist --no-cpu-baseline --settle-seconds 0 --steps 4 --warmup 1 > $R/gpurun_out/pmc_red_sq2.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_red_sq1 2>&1 | grep closest
python tools/pmc_summary.py gpurun_out/pmc_red_sq2 2>&1 | grep closest
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
