# round 6, session 44: selfdist contacts through the rectangular kernels (tiles rotated over the XCDs, rounds model over computing blocks): tests, A-B
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -3)
for rep in 1 2; do
  MKAMD_ALLOW_DIAGNOSTICS=1 MKAMD_LIB=$GRAFT_REPO_ROOT/.variants/libmkamd_prev.so timeout 600 python tools/pair_walk_ab.py 2>&1 | grep -v amdgpu
  timeout 600 python tools/pair_walk_ab.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/pair_walk_ab2.txt
