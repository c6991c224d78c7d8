#!/bin/bash
# tools/build_variant.sh name [hipcc flags...] -- a variant build of the library into .variants/libmkamd_<name>.so
# (git-ignored, travels to the GPU box with the gpurun snapshot; selected with MKAMD_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
n=$1; shift
mkdir -p $R/.variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
    $R/moleculekit_amd/csrc/capi.hip -o $R/.variants/libmkamd_$n.so
echo built $R/.variants/libmkamd_$n.so
