# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 14: the batch's "all roots ordinary" test as an unsigned min / max tree, and the frame kernel in ~1 280 blocks, against the
# committed build (.variants/libmkamd_base.so), alternating
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/.variants/libmkamd_base.so
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x > gpurun_out/pytest_gpu_dist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_dist.log)
tail -3 gpurun_out/pytest_gpu_dist.log
rm -f gpurun_out/dist_ab5.txt
for r in 1 2 3; do
  (PROBE_AVOID=0 PROBE_ODD=1 MKAMD_LIB=$V MKAMD_ALLOW_DIAGNOSTICS=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "avoid=" | sed 's/^/base /') >> gpurun_out/dist_ab5.txt
  (PROBE_AVOID=0 PROBE_ODD=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "avoid=" | sed 's/^/new  /') >> gpurun_out/dist_ab5.txt
done
sort -k2,2n -k4,4n -k5,5 -k6,6 -s gpurun_out/dist_ab5.txt | cut -c1-100
