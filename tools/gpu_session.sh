# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 10: A/B of the image shift as one fma per axis (|r| <= 2), variant .variants/libmkamd_ab1.so against the library
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/.variants/libmkamd_ab1.so
(MKAMD_LIB=$V MKAMD_ALLOW_DIAGNOSTICS=1 timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x > gpurun_out/pytest_gpu_dist_ab1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_dist_ab1.log)
tail -4 gpurun_out/pytest_gpu_dist_ab1.log
for r in 1 2 3; do
  (PROBE_AVOID=0 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "pbc=True" | sed 's/^/base /') >> gpurun_out/dist_ab1.txt
  (PROBE_AVOID=0 MKAMD_LIB=$V MKAMD_ALLOW_DIAGNOSTICS=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "pbc=True" | sed 's/^/ab1  /') >> gpurun_out/dist_ab1.txt
done
sort -k2,2n -k4,4n -s gpurun_out/dist_ab1.txt | cut -c1-110
