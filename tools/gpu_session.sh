# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 12: the pair-table kernel with direct stores (no turn through LDS; mask bit 16, a trial) against the LDS form, same process
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x > gpurun_out/pytest_gpu_dist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_dist.log)
tail -4 gpurun_out/pytest_gpu_dist.log
timeout 120 python tools/_direct_check.py 2>&1 | tail -8
rm -f gpurun_out/dist_ab3.txt
for r in 1 2; do
  (PROBE_AVOID=0,16 PROBE_ODD=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "avoid=") >> gpurun_out/dist_ab3.txt
done
sort -k1,1n -k5,5 -k6,6 -s gpurun_out/dist_ab3.txt | cut -c1-118
