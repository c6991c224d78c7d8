# tools/gpu_r3_latency.sh -- round 3: where one call's microseconds go (cfg2 item and the 3PTB pocket, one grid per call)
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 120 tools/gridsync > gpurun_out/gridsync.txt 2>&1)
(timeout 200 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1); tail -20 gpurun_out/latency_probe.txt
for w in cfg2 3ptb; do rm -rf gpurun_out/st_$w; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st_$w -- python $R/tools/single_timeline.py $w > $R/gpurun_out/st_$w.log 2>&1); python tools/single_timeline_report.py gpurun_out/st_$w > gpurun_out/single_timeline_$w.txt 2>&1; cat gpurun_out/single_timeline_$w.txt; done
if [ -f .variants/libmkamd_phase.so ]; then
  (MKAMD_LIB=.variants/libmkamd_phase.so python tools/phase_timers.py cfg2 1 > gpurun_out/phase_timers_b1.txt 2>&1; MKAMD_LIB=.variants/libmkamd_phase.so python tools/phase_timers.py cfg1 1 >> gpurun_out/phase_timers_b1.txt 2>&1); cat gpurun_out/phase_timers_b1.txt
fi
