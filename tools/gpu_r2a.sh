# tools/gpu_r2a.sh -- round-2 GPU session A: smoke, parity tests, default bench line (with the secondary workloads),
# the RCCL path with one rank, rocprofv3 kernel trace of the headline workload.  Outputs -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(timeout 600 python bench.py > gpurun_out/bench_cfg2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2.log)
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/bench_torchrun1.log 2>&1; echo "rc=$?" >> gpurun_out/bench_torchrun1.log)
(timeout 400 python bench.py --no-cpu-baseline --no-extra --no-pipeline > gpurun_out/bench_cfg2_nopipe.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_nopipe.log)
rm -rf gpurun_out/prof_cfg2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/rocprof_cfg2.log 2>&1; echo "rc=$?" >> $R/gpurun_out/rocprof_cfg2.log)
f=$(find gpurun_out/prof_cfg2 -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 12 2 > gpurun_out/timeline_prof_cfg2.txt 2>&1
(timeout 300 python bench.py --workload dropin --steps 50 > gpurun_out/bench_dropin.log 2>&1; echo "rc=$?" >> gpurun_out/bench_dropin.log)
tail -3 gpurun_out/smoke.log; tail -6 gpurun_out/pytest_gpu.log
for f in gpurun_out/bench_*.log; do echo "== $f"; grep -E "^\{|rc=|Error|error" $f | cut -c1-1800; done
s=$(find gpurun_out/prof_cfg2 -name "*kernel_stats.csv" | head -1); head -12 $s
