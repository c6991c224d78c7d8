# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 6, session 2: 8-wave blocks for the closest-atom reduction, contact masks kept from the count pass, k_dist_pairs<PBC>;
# GPU tier, the dist line, and counters of the reduction leg
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > gpurun_out/s2_all_tests.txt 2>&1
cat gpurun_out/s2_all_tests.txt
(timeout 600 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/s2_bench_dist.log 2>&1; echo "rc=$?" >> gpurun_out/s2_bench_dist.log)
python - <<'PY'
import json
for l in open("gpurun_out/s2_bench_dist.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("periodic", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("shader_clock_ghz"), "open", d["nonperiodic"]["ms_per_step"], d["nonperiodic"]["roofline"]["frac"])
        print("selfdist", d["selfdist"]); print("small", d["small_call"])
        print("reduction", json.dumps(d["reduction"])); print("contacts", json.dumps(d["contacts"]))
PY
tail -2 gpurun_out/s2_bench_dist.log | cut -c1-300
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH"
rm -rf gpurun_out/pmc_red_*
(cd /tmp && MKAMD_DIST_ONLY=reduction timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/pmc_red_sq1 -- python $R/bench.py --workload dist --no-cpu-baseline --settle-seconds 0 --steps 4 --warmup 1 > $R/gpurun_out/pmc_red_sq1.log 2>&1)
(cd /tmp && MKAMD_DIST_ONLY=reduction timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $R/gpurun_out/pmc_red_sq2 -- python $R/bench.py --workload dist --no-cpu-baseline --settle-seconds 0 --steps 4 --warmup 1 > $R/gpurun_out/pmc_red_sq2.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_red_sq1 2>&1 | tail -30
python tools/pmc_summary.py gpurun_out/pmc_red_sq2 2>&1 | tail -30
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
