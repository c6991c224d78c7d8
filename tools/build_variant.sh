#!/bin/bash
# tools/build_variant.sh name [hipcc flags...] -- a DIAGNOSTICS build of the library into .variants/libmkamd_<name>.so
# (git-ignored, travels to the GPU box with the gpurun snapshot; selected with MKAMD_LIB=... MKAMD_ALLOW_DIAGNOSTICS=1).
# The knobs (csrc/mk_diagnostics.h: -DMK_DIAG=<bits>, -DMK_PHASE_TIMERS, -DMK_BIN_TIMERS) only compile with
# -DMKAMD_DIAGNOSTICS_BUILD, which this script always passes; the library then says DIAGNOSTICS in mkamd_version().
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
n=$1; shift
mkdir -p $R/.variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DMKAMD_DIAGNOSTICS_BUILD "$@" \
    $R/moleculekit_amd/csrc/capi.hip -o $R/.variants/libmkamd_$n.so
echo built $R/.variants/libmkamd_$n.so
