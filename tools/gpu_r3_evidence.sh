# tools/gpu_r3_evidence.sh -- round-3 evidence session on one box: parity tests, the default bench line (as the driver runs
# it) and its in-order twin, rocprofv3 kernel-trace stats of both, PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE in their
# own passes, every launch a full batch), the other workloads' stats, the torchrun 1-rank RCCL path.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_cfg2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2.log)
(timeout 400 python bench.py --no-cpu-baseline --no-extra --no-pipeline > gpurun_out/bench_cfg2_nopipe.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_nopipe.log)
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/bench_torchrun1.log 2>&1; echo "rc=$?" >> gpurun_out/bench_torchrun1.log)
rm -rf gpurun_out/prof_* gpurun_out/pmc_*
PROF="--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2"
# (the kernel-trace passes run 40 timed steps after 5 warm-up calls: under the profiler the first dozen launches of a process
#  are 5-10 % slower than the rest -- clocks -- and a 12-launch average says more about that than about the kernel)
TRACE="--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 40 --warmup 5"
for wl in cfg2 cfg1 cfg3 cfg4 cfg5; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$wl -- python $R/bench.py $TRACE --workload $wl > $R/gpurun_out/rocprof_$wl.log 2>&1)
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2_nopipe -- python $R/bench.py $TRACE --no-pipeline > $R/gpurun_out/rocprof_cfg2_nopipe.log 2>&1)
pmc() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python $R/bench.py $PROF $PMC_EXTRA > $R/gpurun_out/pmc_$name.log 2>&1); }
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
PMC_EXTRA="--no-pipeline"
pmc nopipe_fetch FETCH_SIZE
pmc nopipe_write WRITE_SIZE
PMC_EXTRA="--value-tol 1e-6"
pmc tol_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
# the one-molecule call: per-kernel timeline of one cfg2 grid / one 3PTB pocket per call, latencies, the host side of the drop-in call
(bash tools/gpu_timeline_ab.sh moleculekit_amd/csrc/libmkamd.so > gpurun_out/single_call_timeline.txt 2>&1)
(for i in 1 2 3; do timeout 120 python tools/single_latency.py; done > gpurun_out/single_latency.txt 2>&1)
(timeout 200 python tools/team_probe.py > gpurun_out/team_probe.txt 2>&1)
(timeout 120 tools/gridsync > gpurun_out/gridsync.txt 2>&1)
(timeout 200 python tools/dropin_profile.py > gpurun_out/dropin_profile.txt 2>&1)
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; grep -v amdgpu.ids gpurun_out/single_latency.txt
python tools/summarize.py 2>/dev/null | head -20
python tools/collect_profiles_r3.py
