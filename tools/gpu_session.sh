# tools/gpu_session.sh -- the commands of the CURRENT gpurun session (rewritten from session to session; the evidence session
# that the committed profiles come from is tools/gpu_evidence.sh)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q -x -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
grep -a "cutoff shell" gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
bash tools/gpu_tile_ab.sh plain turned
