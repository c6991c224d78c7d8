"""Build libmkamd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libmkamd.so")
SOURCES = ["capi.hip", "pipeline.h", "kernels.h", "mk_device.h", "dist_kernels.h", "dist_pipeline.h", "xtc_reader.h"]
HEADER = os.path.join(_HERE, "..", "include", "mkamd_voxel.h")
HEADER2 = os.path.join(_HERE, "..", "include", "mkamd_distance.h")
HEADER3 = os.path.join(_HERE, "..", "include", "mkamd_xtc.h")


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    return "hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER, HEADER2, HEADER3]
    stale = (not os.path.exists(LIB)) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)
    if force or stale:
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", os.path.join(CSRC, "capi.hip"), "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
