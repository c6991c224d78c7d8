# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 24: the exactness test's slope in a scalar register (one v_mov per periodic pair fewer in the tile kernels) against the
# committed build, alternating; GPU distance tests with the new build first
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/.variants/libmkamd_base.so
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -2)
rm -f gpurun_out/dist_ab7.txt
for r in 1 2 3; do
  (PROBE_AVOID=0,3 PROBE_ODD=1 MKAMD_LIB=$V MKAMD_ALLOW_DIAGNOSTICS=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "pbc=True" | sed 's/^/base /') >> gpurun_out/dist_ab7.txt
  (PROBE_AVOID=0,3 PROBE_ODD=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "pbc=True" | sed 's/^/new  /') >> gpurun_out/dist_ab7.txt
done
sort -k2,2n -k4,4n -k7,7 -s gpurun_out/dist_ab7.txt | cut -c1-112
