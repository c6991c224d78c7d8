# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 19: the open frame kernel with whole 16-byte LDS reads (ds_read_b128 instead of ds_read_b96) against the committed build
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/.variants/libmkamd_base.so
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -2)
rm -f gpurun_out/dist_ab6.txt
for r in 1 2 3; do
  (PROBE_AVOID=0 PROBE_ODD=1 MKAMD_LIB=$V MKAMD_ALLOW_DIAGNOSTICS=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "avoid=" | sed 's/^/base /') >> gpurun_out/dist_ab6.txt
  (PROBE_AVOID=0 PROBE_ODD=1 timeout 300 python tools/dist_shapes_probe.py 2>&1 | grep "avoid=" | sed 's/^/new  /') >> gpurun_out/dist_ab6.txt
done
sort -k2,2n -k4,4n -k5,5 -k6,6 -s gpurun_out/dist_ab6.txt | cut -c1-100
