# tools/gpu_r2_final.sh -- the round-2 evidence session: everything profiles/r2_* is made from
bash tools/gpu_r2a.sh > gpurun_out/session_a.txt 2>&1
bash tools/gpu_pmc.sh > gpurun_out/session_pmc.txt 2>&1
bash tools/gpu_diag.sh > gpurun_out/diag_breakdown.txt 2>&1
(timeout 600 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1)
(timeout 300 python bench.py --workload dist > gpurun_out/bench_distance.log 2>&1; echo "rc=$?" >> gpurun_out/bench_distance.log)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cfg2_nopipe
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2_nopipe -- python $R/bench.py --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/rocprof_cfg2_nopipe.log 2>&1)
f=$(find gpurun_out/prof_cfg2_nopipe -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f 12 2 > gpurun_out/timeline_prof_cfg2_nopipe.txt 2>&1
for wl in cfg3 cfg4; do rm -rf gpurun_out/prof_$wl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$wl -- python $R/bench.py --no-cpu-baseline --no-extra --workload $wl --steps 5 > $R/gpurun_out/rocprof_$wl.log 2>&1); done
tail -5 gpurun_out/session_a.txt | cut -c1-300; cat gpurun_out/diag_breakdown.txt | tail -7; tail -4 gpurun_out/latency_probe.txt
for wl in cfg1 cfg5; do rm -rf gpurun_out/prof_$wl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$wl -- python $R/bench.py --no-cpu-baseline --no-extra --workload $wl --steps 5 > $R/gpurun_out/rocprof_$wl.log 2>&1); done
(MKAMD_LIB=.variants/libmkamd_phase.so python tools/phase_timers.py cfg2 > gpurun_out/phase_timers.txt 2>&1; MKAMD_LIB=.variants/libmkamd_phase.so python tools/phase_timers.py cfg3 >> gpurun_out/phase_timers.txt 2>&1; MKAMD_LIB=.variants/libmkamd_phase.so python tools/phase_timers.py cfg2 1 >> gpurun_out/phase_timers.txt 2>&1)
for w in cfg2 3ptb; do rm -rf gpurun_out/st_$w; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st_$w -- python $R/tools/single_timeline.py $w > $R/gpurun_out/st_$w.log 2>&1); python tools/single_timeline_report.py gpurun_out/st_$w > gpurun_out/single_timeline_$w.txt 2>&1; done
(timeout 200 python tools/dropin_profile.py > gpurun_out/dropin_profile.txt 2>&1)
(bash tools/gpu_pmc_bin.sh > gpurun_out/pmc_bin.txt 2>&1)
(bash tools/gpu_pmc_dist.sh > gpurun_out/pmc_dist.txt 2>&1)
(bash tools/gpu_kstats_reduction.sh > gpurun_out/kstats_reduction.txt 2>&1)
(MKAMD_LIB=$R/.variants/libmkamd_distdiag.so timeout 200 python bench.py --workload dist --no-cpu-baseline > gpurun_out/bench_distance_store_only.log 2>&1)
(python tools/bench_xtc.py 2>/dev/null | tail -1 > gpurun_out/bench_xtc.txt)
((python tools/bench_host_batch.py; python tools/bench_host_dist.py; python tools/bench_calculate_occupancy.py; python tools/bench_voxelize_trajectory.py) 2>/dev/null | grep -v amdgpu.ids > gpurun_out/host_paths.txt)
(bash tools/gpu_pmc_traffic_wl.sh cfg1 cfg3 cfg4 cfg5 dist > gpurun_out/pmc_traffic_wl.txt 2>&1)
