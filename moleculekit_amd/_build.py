"""Build libmkamd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libmkamd.so")
HOST_LIB = os.path.join(CSRC, "libmkamd_host.so")          # the host entry point alone, plain C++ (no ROCm): build_host()
SOURCES = ["capi.hip", "pipeline.h", "kernels.h", "mk_device.h", "mk_diagnostics.h", "dist_kernels.h", "dist_pipeline.h", "xtc_reader.h", "xtc_gpu.h", "cpu_occupancy.h", "host_pack.h"]
HEADER = os.path.join(_HERE, "..", "include", "mkamd_voxel.h")
HEADER2 = os.path.join(_HERE, "..", "include", "mkamd_distance.h")
HEADER3 = os.path.join(_HERE, "..", "include", "mkamd_xtc.h")


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    return "hipcc"


def source_hash() -> str:
    """sha256 over the sources of the library (names and contents), first 16 hex digits: compiled into mkamd_version(), so that
    a measurement can name the build it was taken on (profiles/*_pmc_counters.json; bench.py compares)."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + [HEADER, HEADER2, HEADER3]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash(lib: str = LIB):
    """The source hash a built library carries (None: missing / unstamped)."""
    import re
    try:
        blob = open(lib, "rb").read()
    except OSError:
        return None
    m = re.search(rb"moleculekit_amd [0-9.]+ \(gfx950, HIP\) src ([0-9a-f]{16})", blob)
    return m.group(1).decode() if m else None


def build(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER, HEADER2, HEADER3]
    stale = (not os.path.exists(LIB)) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)
    # (a checkout resets mtimes: the stamp in the library decides then -- same sources, no rebuild)
    if stale and not force and built_hash() == source_hash():
        os.utime(LIB, None)
        stale = False
    if force or stale:
        # hipcc reads the sources twice (device pass, then host pass): a header edited in between gives a library whose
        # host stubs and device code disagree.  Build beside the target, stamp it with the time the compile STARTED
        # (so an edit made during it leaves the library stale) and move it into place in one step.
        started = time.time()
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", '-DMKAMD_SRC_HASH="%s"' % source_hash(), os.path.join(CSRC, "capi.hip"), "-o", tmp]
        # a release build: none of the diagnostics knobs of csrc/mk_diagnostics.h (they compile parts of the kernels out)
        assert not any(a.startswith("-DMK_") or a.startswith("-DMKAMD_DIAG") for a in cmd), "release builds define no MK_* knobs"
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


def build_host(force: bool = False, verbose: bool = False) -> str:
    """libmkamd_host.so: mkamd_calculate_occupancy_cpu[_threads] by the plain C++ compiler ($CXX, g++, c++ or clang++ -- no hipcc,
    nothing of ROCm linked), with -ffp-contract=off: what `occupancy_utils.calculate_occupancy_cpu` / method="CPU" load."""
    import shutil
    deps = [os.path.join(CSRC, "host_capi.cpp"), os.path.join(CSRC, "cpu_occupancy.h")]
    stale = (not os.path.exists(HOST_LIB)) or any(os.path.getmtime(d) > os.path.getmtime(HOST_LIB) for d in deps)
    if force or stale:
        cxx = next((c for c in (os.environ.get("CXX"), "g++", "c++", "clang++") if c and shutil.which(c)), None)
        if cxx is None:
            raise RuntimeError("moleculekit_amd: no C++ compiler found for libmkamd_host.so (set CXX)")
        tmp = "%s.%d.tmp" % (HOST_LIB, os.getpid())
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wall", deps[0], "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, HOST_LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
