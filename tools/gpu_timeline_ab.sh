# tools/gpu_timeline_ab.sh LIB... -- per-kernel timeline of one cfg2 grid / one 3PTB pocket per call for several library builds
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for l in "$@"; do
  n=$(basename $l .so)
  for w in cfg2 3ptb; do rm -rf gpurun_out/tl_${n}_$w; (cd /tmp && MKAMD_LIB=$R/$l timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_${n}_$w -- python $R/tools/single_timeline.py $w > $R/gpurun_out/tl_${n}_$w.log 2>&1); echo "== $n $w"; python tools/single_timeline_report.py gpurun_out/tl_${n}_$w 2>&1 | tail -5; done
done
