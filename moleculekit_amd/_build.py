"""Build libmkamd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import time

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
        # hipcc reads the sources twice (device pass, then host pass): a header edited in between gives a library whose
        # host stubs and device code disagree.  Build beside the target, stamp it with the time the compile STARTED
        # (so an edit made during it leaves the library stale) and move it into place in one step.
        started = time.time()
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", os.path.join(CSRC, "capi.hip"), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.utime(tmp, (started, started))
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
