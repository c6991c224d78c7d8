# round 6, session 38: get_collisions (one frame) at system-building sizes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/collisions_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/collisions_probe.txt
MKAMD_ALLOW_DIAGNOSTICS=1 MKAMD_LIB=$GRAFT_REPO_ROOT/.variants/libmkamd_prev.so timeout 900 python tools/collisions_probe.py 2>&1 | grep -v amdgpu | sed 's/^/prev: /' | tee -a gpurun_out/collisions_probe.txt
