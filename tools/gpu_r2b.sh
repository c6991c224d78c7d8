# tools/gpu_r2b.sh -- round-2 GPU session B: team-of-waves tile kernel (parity + latency), trace of single-grid calls
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "team" > gpurun_out/pytest_team.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_team.log)
(timeout 600 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1; echo "rc=$?" >> gpurun_out/latency_probe.txt)
rm -rf gpurun_out/prof_single
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/prof_single -- python $R/bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/rocprof_single.log 2>&1; echo "rc=$?" >> $R/gpurun_out/rocprof_single.log)
f=$(find gpurun_out/prof_single -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 40 0 > gpurun_out/timeline_single.txt 2>&1
tail -5 gpurun_out/pytest_team.log; cat gpurun_out/latency_probe.txt; tail -45 gpurun_out/timeline_single.txt
