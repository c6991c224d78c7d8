"""tests/emu_build.py -- TEST INFRASTRUCTURE: build + ctypes front-end for tests/emu/libmkamd_emu.so.

The emulation library compiles the product's kernel source (moleculekit_amd/csrc/kernels.h) and
launch sequences (pipeline.h) for the HOST with fibers standing in for GPU lanes, so the CPU-only
test tier can check the kernels' logic against the oracle.  It is never imported by the product.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = os.path.join(_HERE, "emu")
_LIB = os.path.join(_EMU, "libmkamd_emu.so")
_CSRC = os.path.join(_HERE, "..", "moleculekit_amd", "csrc")
_lib = None


def build(force=False):
    srcs = [os.path.join(_EMU, "emu_capi.cpp"), os.path.join(_EMU, "emu_device.h"),
            os.path.join(_CSRC, "kernels.h"), os.path.join(_CSRC, "pipeline.h")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(
            ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
             "-Wno-unused-variable", "-Wno-unknown-pragmas", "-ffp-contract=off",
             os.path.join(_EMU, "emu_capi.cpp"), "-o", _LIB])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.emu_last_error.restype = ctypes.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def voxelize_lattice(coords, atom_offsets, sigmas, origins, nvox, voxelsize, box=None, max_images=0,
                     tile_k=0, force_general=False):
    coords = np.ascontiguousarray(coords, np.float32).reshape(-1, 3)
    atom_offsets = np.ascontiguousarray(atom_offsets, np.int64)
    sig64 = sigmas.dtype == np.float64
    sigmas = np.ascontiguousarray(sigmas, np.float64 if sig64 else np.float32)
    origins = np.ascontiguousarray(origins, np.float64).reshape(-1, 3)
    nvox = np.ascontiguousarray(nvox, np.int32)
    B, C = origins.shape[0], sigmas.shape[1]
    V = int(np.prod(nvox))
    out = np.empty((B, V, C), np.float32)
    bx = None if box is None else np.ascontiguousarray(box, np.float32).reshape(B, 3)
    err = ctypes.c_int(0)
    st = lib().emu_voxelize_lattice(
        ctypes.c_int(B), _p(coords), _p(atom_offsets), _p(sigmas), ctypes.c_int(int(sig64)), ctypes.c_int(C),
        _p(origins), _p(nvox), ctypes.c_double(voxelsize), _p(bx), ctypes.c_int(max_images),
        ctypes.c_int(tile_k), ctypes.c_int(int(force_general)), _p(out), ctypes.byref(err))
    if st != 0:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return out, err.value


def occupancy_centers(centers, coords, sigmas, box=None):
    centers = np.ascontiguousarray(centers, np.float64)
    coords = np.ascontiguousarray(coords, np.float32)
    sig64 = sigmas.dtype == np.float64
    sigmas = np.ascontiguousarray(sigmas, np.float64 if sig64 else np.float32)
    V, N, C = centers.shape[0], coords.shape[0], sigmas.shape[1]
    out = np.empty((V, C), np.float32)
    bx = None if box is None else np.ascontiguousarray(box, np.float64).reshape(3)
    st = lib().emu_occupancy_centers(_p(centers), ctypes.c_longlong(V), _p(coords), ctypes.c_longlong(N),
                                     _p(sigmas), ctypes.c_int(int(sig64)), ctypes.c_int(C), _p(bx), _p(out))
    if st != 0:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return out


def grid_centers(bb_min, nvox, voxelsize):
    bb_min = np.ascontiguousarray(bb_min, np.float64)
    nvox = np.ascontiguousarray(nvox, np.int32)
    out = np.empty((int(np.prod(nvox)), 3), np.float64)
    st = lib().emu_grid_centers(_p(bb_min), _p(nvox), ctypes.c_double(voxelsize), _p(out))
    assert st == 0
    return out


def exclusive_scan(counts):
    counts = np.ascontiguousarray(counts, np.uint32)
    out = np.empty(counts.size + 1, np.uint32)
    st = lib().emu_exclusive_scan(_p(counts), ctypes.c_longlong(counts.size), _p(out))
    assert st == 0
    return out


def plan(B, total_atoms, C, nvox, voxelsize, pbc=0, max_images=1, tile_k=0):
    nvox = np.ascontiguousarray(nvox, np.int32)
    out = np.zeros(16, np.int32)
    st = lib().emu_plan(ctypes.c_int(B), ctypes.c_longlong(total_atoms), ctypes.c_int(C), _p(nvox),
                        ctypes.c_double(voxelsize), ctypes.c_int(pbc), ctypes.c_int(max_images),
                        ctypes.c_int(tile_k), _p(out))
    if st != 0:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    keys = ["K", "tnx", "tny", "tnz", "ntiles", "cs", "h", "ncx", "ncy", "ncz", "ncell", "rint", "G", "M"]
    return dict(zip(keys, out.tolist()))
