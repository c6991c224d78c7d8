# round 6, session 15: the reduction kernel held to 96 / 84 registers (five / six waves per SIMD) against the release build
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2; do
  PROBE_BLOCKS=0 timeout 300 python tools/reduction_probe.py 2>&1 | grep -v amdgpu | head -3 | cut -c1-200 | sed 's/^/release /'
  for v in vgpr96 vgpr84; do
    MKAMD_LIB=$PWD/.variants/libmkamd_$v.so MKAMD_ALLOW_DIAGNOSTICS=1 PROBE_BLOCKS=0 timeout 300 python tools/reduction_probe.py 2>&1 | grep -v amdgpu | head -3 | cut -c1-200 | sed "s/^/$v /"
  done
done
