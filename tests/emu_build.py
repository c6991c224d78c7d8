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
            os.path.join(_CSRC, "kernels.h"), os.path.join(_CSRC, "pipeline.h"),
            os.path.join(_CSRC, "dist_kernels.h"), os.path.join(_CSRC, "dist_pipeline.h"), os.path.join(_CSRC, "xtc_gpu.h"),
            os.path.join(_CSRC, "host_pack.h")]
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
        _lib.emu_last_dist_kernel.restype = ctypes.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def voxelize_lattice(coords, atom_offsets, sigmas, origins, nvox, voxelsize, box=None, max_images=0,
                     tile_k=0, force_general=False, affine=None, lds_tier=-1, feedback=None, prepass_mode=-1, tile_team=-1, fine_cells=False,
                     repeat=1, fills=None, tile_items=-1, value_tol=0.0, direct=-1, cell_cap=0, spill_cap=0, direct_words=None):
    """feedback: optional uint32[4] array, in = tier statistics of the 'previous call', out = this call's.
    repeat: run the call that many times on ONE backend (workspace kept) and return the last result; fills: optional
    one-element list that receives the number of counter memsets those calls issued."""
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
    nfills = ctypes.c_int(0)
    st = lib().emu_voxelize_lattice(
        ctypes.c_int(B), _p(coords), _p(atom_offsets), _p(sigmas), ctypes.c_int(int(sig64)), ctypes.c_int(C),
        _p(origins), _p(nvox), ctypes.c_double(voxelsize), _p(bx), ctypes.c_int(max_images),
        ctypes.c_int(tile_k), ctypes.c_int(int(force_general)),
        _p(None if affine is None else np.ascontiguousarray(affine, np.float64)), _p(out), ctypes.byref(err),
        ctypes.c_int(lds_tier), _p(feedback), ctypes.c_int(prepass_mode), ctypes.c_int(tile_team), ctypes.c_int(int(fine_cells)),
        ctypes.c_int(int(repeat)), ctypes.byref(nfills), ctypes.c_int(int(tile_items)), ctypes.c_double(float(value_tol)),
        ctypes.c_int(int(direct)), ctypes.c_int(int(cell_cap)), ctypes.c_uint(int(spill_cap)), _p(direct_words))
    if fills is not None:
        fills[:] = [nfills.value]
    if st != 0:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return out, err.value


def voxelize_lattice_topo(coords, sigmas_one, n_items, origins, nvox, voxelsize, box=None, max_images=0, tile_k=0, affine=None, repeat=1,
                          atom_offsets=None, exact_redo=0):
    """B sets of coordinates of ONE molecule (sigmas_one [n, C]) through the topology path (run_topology_build + run_lattice with
    P.topo) on the emulated kernels -> (features [B, V, C], device error flag, the handle's wide flag)."""
    coords = np.ascontiguousarray(coords, np.float32).reshape(-1, 3)
    sig64 = sigmas_one.dtype == np.float64
    sigmas_one = np.ascontiguousarray(sigmas_one, np.float64 if sig64 else np.float32)
    n, C = sigmas_one.shape
    B = int(n_items)
    offs = np.ascontiguousarray(np.arange(B + 1, dtype=np.int64) * n if atom_offsets is None else atom_offsets, np.int64)
    origins = np.ascontiguousarray(origins, np.float64).reshape(B, 3)
    nvox = np.ascontiguousarray(nvox, np.int32)
    out = np.empty((B, int(np.prod(nvox)), C), np.float32)
    bx = None if box is None else np.ascontiguousarray(box, np.float32).reshape(B, 3)
    err, wide = ctypes.c_int(0), ctypes.c_int(0)
    st = lib().emu_voxelize_lattice_topo(ctypes.c_int(B), _p(coords), _p(offs), _p(sigmas_one), ctypes.c_int(int(sig64)), ctypes.c_longlong(n),
                                         ctypes.c_int(C), _p(origins), _p(nvox), ctypes.c_double(voxelsize), _p(bx), ctypes.c_int(max_images),
                                         ctypes.c_int(tile_k), _p(None if affine is None else np.ascontiguousarray(affine, np.float64)), _p(out),
                                         ctypes.byref(err), ctypes.byref(wide), ctypes.c_int(int(repeat)), ctypes.c_int(int(exact_redo)))
    if st != 0:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return out, err.value, bool(wide.value)


def choose_tier(forced, feedback):
    return int(lib().emu_choose_tier(ctypes.c_int(forced), _p(np.ascontiguousarray(feedback, np.uint32))))


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


# ---- distance_utils row ---------------------------------------------------------------------------
def _csr(groups):
    offs = np.zeros(len(groups) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(g) for g in groups])
    return np.array([a for g in groups for a in g], dtype=np.int32), offs


def _npairs(n1, n2, selfdist):
    return sum(max(n2 - 1 - i, 0) for i in range(n1)) if selfdist else n1 * n2


def dist_trajectory(coords, box, sel1, sel2, chains, selfdist, pbc, squared=False, avoid=0):
    """avoid: DIST_AVOID_* bits of dist_pipeline.h (1 the block-per-frame kernel, 2 the row kernel, 4 the rectangular tile kernel,
    8 the row kernel's 16-byte stores): the call takes the best kernel among the rest"""
    coords = np.ascontiguousarray(coords, np.float32); box = np.ascontiguousarray(box, np.float32)
    sel1 = np.ascontiguousarray(sel1, np.uint32); sel2 = np.ascontiguousarray(sel2, np.uint32)
    chains = np.ascontiguousarray(chains, np.uint32)
    F = coords.shape[2]
    out = np.full((F, _npairs(len(sel1), len(sel2), selfdist)), -7.0, np.float32)
    st = lib().emu_dist_trajectory(_p(coords), ctypes.c_longlong(F), _p(box), _p(sel1), ctypes.c_longlong(len(sel1)), _p(sel2),
                                   ctypes.c_longlong(len(sel2)), _p(chains), ctypes.c_int(int(selfdist)), ctypes.c_int(int(pbc)),
                                   ctypes.c_int(int(squared)), _p(out), ctypes.c_int(int(avoid)))
    assert st == 0, lib().emu_last_error()
    return out


def last_dist_kernel():
    """The kernels the last dist_trajectory call chose (run_dist_trajectory's note, as mkamd_ctx_last_dist_kernel reports it on the device)."""
    return lib().emu_last_dist_kernel().decode()


def dist_reduction(coords, box, groups1, groups2, ch1, ch2, selfdist, pbc, masses, r1, r2, pairs=False, block=0, n_atoms=None):
    """block: 0 = the library's choice, 4 / 8 = first-group atoms in registers (closest / closest only), -1 = the generic kernel,
    -2 = the few-frame kernel (lanes along the second groups) at any number of frames;
    n_atoms: what the host believes the number of coordinate rows is (decides 32-bit row offsets: a huge value forces the 64-bit form)."""
    coords = np.ascontiguousarray(coords, np.float32); box = np.ascontiguousarray(box, np.float32)
    a1, o1 = _csr(groups1); a2, o2 = _csr(groups2)
    ch1 = np.ascontiguousarray(ch1, np.uint32); ch2 = np.ascontiguousarray(ch2, np.uint32)
    masses = np.ascontiguousarray(masses, np.float32)
    F = coords.shape[2]
    nout = len(groups1) if pairs else _npairs(len(groups1), len(groups2), selfdist)
    out = np.full((F, nout), -7.0, np.float32)
    st = lib().emu_dist_reduction(_p(coords), ctypes.c_longlong(F), _p(box), _p(a1), _p(o1), ctypes.c_longlong(len(groups1)),
                                  _p(a2), _p(o2), ctypes.c_longlong(len(groups2)), _p(ch1), _p(ch2), ctypes.c_int(int(selfdist)),
                                  ctypes.c_int(int(pairs)), ctypes.c_int(int(pbc)), _p(masses), ctypes.c_int(r1), ctypes.c_int(r2), _p(out),
                                  ctypes.c_longlong(coords.shape[0] if n_atoms is None else n_atoms), ctypes.c_int(int(block)))
    assert st == 0, lib().emu_last_error()
    return out


def contacts_trajectory(coords, box, sel1, sel2, chains, selfdist, pbc, threshold, budget_bytes=256 << 20, device_sink=False, avoid=0):
    """-> per-frame list of flat [a0, b0, a1, b1, ...] lists (the reference's return shape), through the device-side
    count / scan / fill kernels; a tiny `budget_bytes` forces several chunks of frames; `device_sink`: the list is kept in ONE
    growing 'device' buffer as mkamd_contacts_trajectory_dev keeps it (else chunk by chunk to the host, the "_host" form);
    `avoid` = 1: a rectangular call takes the pair-table walk as well (CONTACTS_AVOID_RECT)."""
    coords = np.ascontiguousarray(coords, np.float32); box = np.ascontiguousarray(box, np.float32)
    sel1 = np.ascontiguousarray(sel1, np.uint32); sel2 = np.ascontiguousarray(sel2, np.uint32)
    chains = np.ascontiguousarray(chains, np.uint32)
    F = coords.shape[2]
    cap = F * _npairs(len(sel1), len(sel2), selfdist)
    offs = np.zeros(F + 1, np.int64)
    pairs = np.zeros(2 * max(cap, 1), np.uint32)
    n = ctypes.c_longlong(0)
    st = lib().emu_contacts(_p(coords), ctypes.c_longlong(F), _p(box), _p(sel1), ctypes.c_longlong(len(sel1)), _p(sel2),
                            ctypes.c_longlong(len(sel2)), _p(chains), ctypes.c_int(int(selfdist)), ctypes.c_int(int(pbc)),
                            ctypes.c_float(threshold), ctypes.c_longlong(budget_bytes), _p(offs), _p(pairs), ctypes.c_longlong(cap),
                            ctypes.byref(n), ctypes.c_int(int(device_sink)), ctypes.c_int(int(avoid)))
    assert st == 0, lib().emu_last_error()
    flat = pairs[:2 * n.value].astype(np.int64)
    return [flat[2 * offs[f]:2 * offs[f + 1]].tolist() for f in range(F)]


def pack_atoms(coords, sels, per_atom=None):
    """csrc/host_pack.h (what the library's host entry points of the distance functions do before the upload): -> (on, uniq, packed coords
    [M, 3, F], the selections in the packed numbering, `per_atom` gathered) for selections `sels` (a list of index arrays)."""
    coords = np.ascontiguousarray(coords, np.float32)
    N, _, F = coords.shape
    flat = np.ascontiguousarray(np.concatenate([np.asarray(s, np.uint32).ravel() for s in sels]) if sels else np.zeros(0, np.uint32), np.uint32)
    uniq = np.zeros(max(len(flat), 1), np.uint32)
    out = np.full((max(len(flat), 1), 3, F), np.nan, np.float32)
    remap = np.zeros(max(len(flat), 1), np.uint32)
    M = ctypes.c_longlong(0)
    on = lib().emu_pack_atoms(_p(coords), ctypes.c_longlong(N), ctypes.c_longlong(F), _p(flat), ctypes.c_longlong(len(flat)), _p(out), _p(uniq), _p(remap),
                              ctypes.byref(M))
    m = M.value
    back = remap[:len(flat)].copy()
    if on:
        lib().emu_unpack_atoms(_p(uniq), ctypes.c_longlong(m), _p(back), ctypes.c_longlong(len(back)))
    return bool(on), uniq[:m].copy(), out[:m].copy(), remap[:len(flat)].copy(), back


def cdist(c1, c2):
    c1 = np.ascontiguousarray(c1, np.float32); c2 = np.ascontiguousarray(c2, np.float32)
    out = np.full((c1.shape[0], c2.shape[0]), -7.0, np.float32)
    assert lib().emu_cdist(_p(c1), ctypes.c_longlong(c1.shape[0]), _p(c2), ctypes.c_longlong(c2.shape[0]), ctypes.c_int(c1.shape[1]), _p(out)) == 0
    return out


def pdist(c):
    c = np.ascontiguousarray(c, np.float32)
    n = c.shape[0]
    out = np.full(n * (n - 1) // 2, -7.0, np.float32)
    assert lib().emu_pdist(_p(c), ctypes.c_longlong(n), ctypes.c_int(c.shape[1]), _p(out)) == 0
    return out


def calculate_occupancy(centers, coords, sigmas, results, inject_lattice_status=0):
    """mkamd_calculate_occupancy's routing (lattice recogniser + route_calculate_occupancy of pipeline.h) on the emulated
    kernels: max-accumulates into `results` (float64 [V, C]) and returns (status, route) with route 1 = tiled lattice
    kernels, 2 = pairwise kernel, 0 = an error came back."""
    centers = np.ascontiguousarray(centers, np.float64).reshape(-1, 3)
    coords = np.ascontiguousarray(coords, np.float32).reshape(-1, 3)
    sigmas = np.ascontiguousarray(sigmas, np.float64)
    assert results.dtype == np.float64 and results.flags["C_CONTIGUOUS"]
    route = ctypes.c_int(0)
    st = lib().emu_calculate_occupancy(_p(centers), ctypes.c_longlong(centers.shape[0]), _p(coords),
                                       ctypes.c_longlong(coords.shape[0]), _p(sigmas), ctypes.c_int(sigmas.shape[1]),
                                       _p(results), ctypes.c_int(inject_lattice_status), ctypes.byref(route))
    return st, route.value


def xtc_decode(raw, desc, natoms, scale=1.0):
    """The device XTC decoder's two kernels (csrc/xtc_gpu.h) on host memory: ``raw`` uint8 (the records + XTC_PAD bytes),
    ``desc`` uint8 [n, 64] -> (xyz float32 [n, natoms, 3] (poisoned with NaN first), status int32 [n])."""
    raw = np.ascontiguousarray(raw, np.uint8)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = desc.shape[0]
    xyz = np.full((n, natoms, 3), np.nan, np.float32)
    status = np.full(n, -1, np.int32)
    rc = lib().emu_xtc_decode(_p(raw), _p(desc), ctypes.c_longlong(n), ctypes.c_longlong(natoms), ctypes.c_float(scale), _p(xyz), _p(status))
    assert rc == 0
    return xyz, status

