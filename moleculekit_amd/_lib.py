"""ctypes binding of ``libmkamd.so`` -- the C ABI declared in ``include/mkamd_voxel.h``.

The library holds the hand-written HIP kernels (gfx950) of the voxel-descriptor hot path.  There
is NO CPU compute path behind this module: if the shared library is missing, or no MI355X/HIP
device is visible, every compute entry point raises ``RuntimeError`` -- loudly.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MKAMD_LIB", os.path.join(_HERE, "csrc", "libmkamd.so"))

MKAMD_OK, MKAMD_EINVAL, MKAMD_EHIP, MKAMD_ENODEV, MKAMD_EOVERFLOW, MKAMD_EBOX, MKAMD_ENOMEM = range(7)

_c_int, _c_i32, _c_i64, _c_dbl, _vp = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes): every symbol include/mkamd_voxel.h declares
SIGNATURES = {
    "mkamd_version": (ctypes.c_char_p, []),
    "mkamd_last_error": (ctypes.c_char_p, []),
    "mkamd_device_count": (_c_int, [ctypes.POINTER(_c_int)]),
    "mkamd_ctx_create": (_c_int, [_c_int, ctypes.POINTER(_vp)]),
    "mkamd_ctx_destroy": (_c_int, [_vp]),
    "mkamd_ctx_set_stream": (_c_int, [_vp, _vp]),
    "mkamd_ctx_synchronize": (_c_int, [_vp]),
    "mkamd_ctx_abandon_pending": (_c_int, [_vp]),
    "mkamd_ctx_poll_errors": (_c_int, [_vp]),
    "mkamd_ctx_device_info": (_c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_c_int),
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_size_t]),
    "mkamd_ctx_set_tile_k": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_force_general": (_c_int, [_vp, _c_int]),
    "mkamd_xtc_info": (_c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "mkamd_xtc_read": (_c_int, [ctypes.c_char_p, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _vp, ctypes.c_int32]),
    "mkamd_xtc_chunk_desc": (_c_int, [ctypes.c_char_p, _vp, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int64), _vp, _vp, _vp]),
    "mkamd_xtc_copy_bytes": (_c_int, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int32]),
    "mkamd_xtc_byte_range": (_c_int, [ctypes.c_char_p, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "mkamd_xtc_chunk_desc_mem": (_c_int, [ctypes.c_char_p, _vp, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int64, _vp,
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _vp, _vp, _vp]),
    "mkamd_xtc_decode_work_bytes": (ctypes.c_uint64, [ctypes.c_int64, ctypes.c_int64]),
    "mkamd_xtc_decode_dev": (_c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, _vp, _vp, _vp, ctypes.c_uint64]),
    "mkamd_ctx_set_lds_tier": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_prepass_mode": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_pipelining": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_promise_inputs": (_c_int, [_vp, _vp]),
    "mkamd_ctx_withdraw_promise": (_c_int, [_vp]),
    "mkamd_ctx_pipelined_calls": (_c_int, [_vp, ctypes.POINTER(_c_i64)]),
    "mkamd_ctx_last_tile_kernel": (_c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "mkamd_clock_probe_dev": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "mkamd_frames_to_items_dev": (_c_int, [_vp, _vp, _vp, _c_i64, _c_i64, _c_i64, ctypes.c_float, _vp]),
    "mkamd_ctx_set_tile_team": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_tile_items": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_exact_redo": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_fine_cells": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_set_value_tolerance": (_c_int, [_vp, ctypes.c_double]),
    "mkamd_ctx_set_direct_binning": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_enable_kernel_timing": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_read_kernel_timing": (_c_int, [_vp, ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_i64)]),
    "mkamd_calculate_occupancy": (_c_int, [_vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_i32, _vp]),
    "mkamd_calculate_occupancy_cpu": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_i32, _vp]),
    "mkamd_calculate_occupancy_cpu_threads": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_i32, _vp, _c_i32]),
    "mkamd_occupancy_centers_host": (_c_int, [_vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_int, _c_i32, _vp, _vp]),
    "mkamd_occupancy_centers_dev": (_c_int, [_vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_int, _c_i32, _vp, _vp]),
    "mkamd_voxelize_lattice_host": (_c_int, [_vp, _c_i32, _vp, _vp, _vp, _c_int, _c_i32, _vp, _vp, _c_dbl,
                                             _vp, _c_i32, _vp]),
    "mkamd_voxelize_lattice_host_f64": (_c_int, [_vp, _c_i32, _vp, _vp, _vp, _c_int, _c_i32, _vp, _vp, _c_dbl,
                                                 _vp, _c_i32, _vp]),
    "mkamd_voxelize_lattice_host_begin": (_c_int, [_vp, _c_i32, _vp, _vp, _vp, _c_int, _c_i32, _vp, _vp, _c_dbl, _vp, _c_i32]),
    "mkamd_voxelize_lattice_host_end": (_c_int, [_vp, _vp, _vp, ctypes.c_uint64]),
    "mkamd_voxelize_lattice_dev": (_c_int, [_vp, _c_i32, _vp, _vp, _c_i64, _vp, _c_int, _c_i32, _vp, _vp,
                                            _c_dbl, _vp, _c_i32, _vp]),
    "mkamd_voxelize_lattice_aug_dev": (_c_int, [_vp, _c_i32, _vp, _vp, _c_i64, _vp, _c_int, _c_i32, _vp, _vp,
                                                _c_dbl, _vp, _c_i32, _vp, _vp]),
    "mkamd_topology_create_dev": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_i32, _c_dbl, ctypes.POINTER(_vp)]),
    "mkamd_topology_create_host": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_i32, _c_dbl, ctypes.POINTER(_vp)]),
    "mkamd_topology_destroy": (_c_int, [_vp, _vp]),
    "mkamd_topology_info": (_c_int, [_vp, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i32), ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_i32)]),
    "mkamd_voxelize_lattice_topo_dev": (_c_int, [_vp, _c_i32, _vp, _vp, _c_i64, _vp, _vp, _vp, _c_dbl, _vp, _c_i32, _vp, _vp]),
    "mkamd_grid_centers_host": (_c_int, [_vp, _vp, _vp, _c_dbl, _vp]),
    "mkamd_grid_centers_dev": (_c_int, [_vp, _vp, _vp, _c_dbl, _vp]),
    "mkamd_lattice_from_centers": (_c_int, [_vp, ctypes.c_int64, _vp, _vp, _vp]),
    "mkamd_prefault": (_c_int, [_vp, ctypes.c_uint64]),
    "mkamd_copy_to_host": (_c_int, [_vp, _vp, _vp, ctypes.c_uint64]),
    "mkamd_copy_dev": (_c_int, [_vp, _vp, _vp, ctypes.c_uint64]),
    # include/mkamd_distance.h
    "mkamd_dist_count_pairs": (_c_i64, [_c_i64, _c_i64, _c_int]),
    "mkamd_dist_trajectory_host": (_c_int, [_vp, _vp, _c_i64, _c_i64, _vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_int, _c_int,
                                            _c_int, _vp]),
    "mkamd_dist_trajectory_dev": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_int, _c_int, _c_int, _vp]),
    "mkamd_contacts_trajectory_host": (_c_int, [_vp, _vp, _c_i64, _c_i64, _vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_int, _c_int,
                                                ctypes.c_float, _vp, ctypes.POINTER(_vp)]),
    "mkamd_dist_reduction_host": (_c_int, [_vp, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _vp, _vp,
                                           _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp]),
    "mkamd_contacts_trajectory_dev": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_int, _c_int, ctypes.c_float, _vp,
                                               ctypes.POINTER(_vp)]),
    "mkamd_dist_reduction_dev": (_c_int, [_vp, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _c_i64, _c_i64, _vp, _vp, _c_i64, _vp, _vp,
                                          _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp]),
    "mkamd_cdist_dev": (_c_int, [_vp, _vp, _c_i64, _vp, _c_i64, _c_i32, _vp]),
    "mkamd_pdist_dev": (_c_int, [_vp, _vp, _c_i64, _c_i32, _vp]),
    "mkamd_ctx_set_reduction_block": (_c_int, [_vp, _c_int]),
    "mkamd_selftest_sqrt": (_c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]),
    "mkamd_ctx_set_dist_kernels": (_c_int, [_vp, _c_int]),
    "mkamd_ctx_last_dist_kernel": (_c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "mkamd_cdist_host": (_c_int, [_vp, _vp, _c_i64, _vp, _c_i64, _c_i32, _vp]),
    "mkamd_pdist_host": (_c_int, [_vp, _vp, _c_i64, _c_i32, _vp]),
}

_lib = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """Load libmkamd.so (built by ``__graft_entry__.build()`` / ``moleculekit_amd._build``)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"moleculekit_amd: HIP library not found at {LIB_PATH}. Build it with "
                    f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                    f"There is no CPU fallback.")
            if os.environ.get("MKAMD_NO_TORCH_PRELOAD", "0") != "1":
                # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64; if ours
                # (linked against /opt/rocm) initialises first, torch.cuda later reports "No HIP GPUs".
                # Importing torch first makes the loader resolve our DT_NEEDED to the copy torch loaded.
                try:
                    import torch  # noqa: F401
                except Exception:
                    pass
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)          # AttributeError if a declared symbol is missing
                fn.restype, fn.argtypes = res, args
            # a DIAGNOSTICS build (csrc/mk_diagnostics.h: parts of the kernels compiled out to time them -- wrong values on
            # purpose -- or cycle counters inside them) is never picked up by accident
            if b"DIAGNOSTICS" in L.mkamd_version() and os.environ.get("MKAMD_ALLOW_DIAGNOSTICS", "0") != "1":
                raise RuntimeError(f"moleculekit_amd: {LIB_PATH} is a DIAGNOSTICS build ({L.mkamd_version().decode()}); "
                                   f"set MKAMD_ALLOW_DIAGNOSTICS=1 to load it for timing experiments")
            _lib = L
    return _lib


_host_lib = None


def load_host() -> ctypes.CDLL:
    """libmkamd_host.so: the HOST implementation of calculate_occupancy alone (csrc/host_capi.cpp), built on first use by the plain
    C++ compiler -- no hipcc, no HIP runtime: what a machine without ROCm can still load."""
    global _host_lib
    with _lock:
        if _host_lib is None:
            from . import _build
            L = ctypes.CDLL(_build.build_host())
            for name in ("mkamd_calculate_occupancy_cpu", "mkamd_calculate_occupancy_cpu_threads"):
                fn = getattr(L, name)
                fn.restype, fn.argtypes = SIGNATURES[name]
            L.mkamd_host_last_error.restype = ctypes.c_char_p
            _host_lib = L
        return _host_lib


def version() -> str:
    """``mkamd_version()`` of the loaded library: name, version, and the hash of the sources it was built from."""
    return load().mkamd_version().decode()


def source_hash():
    """The 16-hex-digit source hash the loaded library carries (``None`` for a build without a stamp, e.g. a tools/ variant):
    what ties a profile under profiles/ to the build it was taken on."""
    import re
    m = re.search(r"src ([0-9a-f]{16})", version())
    return m.group(1) if m else None


class MkamdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmkamd error {code}: {msg}")
        self.code = code


def _check(st):
    if st != MKAMD_OK:
        msg = load().mkamd_last_error().decode(errors="replace")
        if st == MKAMD_EINVAL:
            raise ValueError(f"libmkamd: {msg}")
        if st == MKAMD_ENOMEM:
            raise MemoryError(f"libmkamd: {msg}")
        raise MkamdError(st, msg)


def device_count() -> int:
    n = _c_int(0)
    st = load().mkamd_device_count(ctypes.byref(n))
    return n.value if st == MKAMD_OK else 0


def _ptr(a):
    """void* of a numpy array / an object with data_ptr() (a torch tensor) / int address / None."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    if isinstance(a, np.ndarray):
        return a.ctypes.data          # plain address: every pointer parameter is declared c_void_p (SIGNATURES); the
                                      # caller's frame keeps the array alive for the duration of the call
    return int(a)


class Context:
    """One device context (``mkamd_ctx``): a HIP stream plus a grow-only workspace."""

    def __init__(self, device: int = 0):
        self._h = _vp(None)
        L = load()
        h = _vp(None)
        _check(L.mkamd_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().mkamd_ctx_destroy(self._h)
            self._h = _vp(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- control -----------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        """hip_stream: integer handle of a hipStream_t (0 = the legacy default stream); None = the context's own stream."""
        h = _vp(-1 & 0xFFFFFFFFFFFFFFFF) if hip_stream is None else _vp(int(hip_stream))
        _check(load().mkamd_ctx_set_stream(self._h, h))

    def synchronize(self):
        _check(load().mkamd_ctx_synchronize(self._h))

    def abandon_pending(self):
        """Give up a host call that was begun and will not be ended (include/mkamd_voxel.h); no-op without one."""
        if getattr(self, "_h", None) is not None and self._h.value:
            _check(load().mkamd_ctx_abandon_pending(self._h))

    def poll_errors(self):
        """Non-blocking: raise if an already finished asynchronous lattice call flagged an error (bad / too small
        periodic box, more images than reserved)."""
        _check(load().mkamd_ctx_poll_errors(self._h))

    def set_tile_k(self, k: int):
        _check(load().mkamd_ctx_set_tile_k(self._h, int(k)))

    def set_lds_tier(self, tier: int):
        """-1 adaptive (default), 0/1/2 = 640/768/1024 LDS entries per tile (include/mkamd_voxel.h)."""
        _check(load().mkamd_ctx_set_lds_tier(self._h, int(tier)))

    def set_prepass_mode(self, mode: int):
        """-1 automatic (default), 0 multi-kernel chain, 1 one-launch per-item pre-pass (include/mkamd_voxel.h)."""
        _check(load().mkamd_ctx_set_prepass_mode(self._h, int(mode)))

    def set_tile_team(self, mode: int):
        """-1 automatic (default), 0 one wave per tile, 1 a team of waves per tile, 4 / 8 / 16 a team of that many
        (include/mkamd_voxel.h)."""
        _check(load().mkamd_ctx_set_tile_team(self._h, int(mode)))

    def set_tile_items(self, mode: int):
        """-1 automatic (default), 0 tiles on their own, 1 a workgroup per item that sorts the item's entries once
        (batches of ligand-sized items; include/mkamd_voxel.h)."""
        _check(load().mkamd_ctx_set_tile_items(self._h, int(mode)))

    def set_exact_redo(self, mode: int = 0):
        """0 (default): a topology call with wide sigmas hands its exact cut-off hits to a launch of their own (many waves per value);
        -1: recomputed inside the call's last launch (include/mkamd_voxel.h).  Same bits either way."""
        _check(load().mkamd_ctx_set_exact_redo(self._h, int(mode)))

    def set_fine_cells(self, on: bool):
        """Half-cutoff cells instead of cutoff-sized ones (A-B benchmarking; same values to float32 noise)."""
        _check(load().mkamd_ctx_set_fine_cells(self._h, int(bool(on))))

    def set_direct_binning(self, mode: int):
        """-1 automatic (default: small calls take the one-launch pre-pass, big calls that are not pipelined the one-pass direct
        binning), 0 the count / scan / fill chain always, 1 the one-pass direct binning for every big call, 2 the one-launch
        pre-pass for any size (include/mkamd_voxel.h); bit-identical results."""
        _check(load().mkamd_ctx_set_direct_binning(self._h, int(mode)))

    def set_value_tolerance(self, eps: float):
        """Tolerance-aware reach (include/mkamd_voxel.h): 0 = off (default, the reference's hard 5 A cutoff), else every
        atom is culled per tile where it is worth less than ``eps`` (<= 1e-5): values move by at most ``eps``."""
        _check(load().mkamd_ctx_set_value_tolerance(self._h, float(eps)))
        self._value_tol = float(eps)

    def set_force_general(self, on: bool):
        _check(load().mkamd_ctx_set_force_general(self._h, int(bool(on))))
        self._force_general = bool(on)

    def set_pipelining(self, on: bool):
        """Overlap the pre-pass of a call with the tile kernel of the previous one (see the header for the contract)."""
        _check(load().mkamd_ctx_set_pipelining(self._h, int(bool(on))))

    def promise_inputs(self, event=None):
        """One-shot promise about the NEXT ``voxelize_lattice_dev`` call (include/mkamd_voxel.h): its inputs are complete
        once ``event`` (a ``torch.cuda.Event`` that has been recorded, a raw ``hipEvent_t``, or None = complete already and
        not produced on this context's stream since the previous call) has completed, and stay untouched until the
        features have been consumed -- the call may then run its pre-pass beside the previous call's tile kernel."""
        h = None if event is None else int(getattr(event, "cuda_event", event))
        _check(load().mkamd_ctx_promise_inputs(self._h, h))

    def withdraw_promise(self):
        """Drop a promise no call has consumed."""
        _check(load().mkamd_ctx_withdraw_promise(self._h))

    def clock_probe_dev(self, stream, microseconds, d_ticks2):
        """One wave on `stream` that spins for `microseconds` and stores {shader clock ticks, 100 MHz reference ticks} (uint64 x 2)
        at the device address `d_ticks2`: the clock the device sustains under whatever runs beside it (include/mkamd_voxel.h)."""
        _check(load().mkamd_clock_probe_dev(self._h, int(stream) or None, int(microseconds), int(d_ticks2)))

    def frames_to_items_dev(self, stream, d_src, rows, src_pitch, n_frames, scale, d_dst):
        """[rows][frames] (frame fastest, pitch ``src_pitch``) -> [frames][rows] * scale on ``stream`` (include/mkamd_voxel.h)."""
        _check(load().mkamd_frames_to_items_dev(self._h, int(stream) or None, int(d_src), int(rows), int(src_pitch), int(n_frames),
                                                float(scale), int(d_dst)))

    def last_tile_kernel(self) -> str:
        """Name of the tile kernel the last lattice call launched, as rocprofv3 prints it ('' before the first call)."""
        buf = ctypes.create_string_buffer(128)
        _check(load().mkamd_ctx_last_tile_kernel(self._h, buf, 128))
        return buf.value.decode()

    def selftest_sqrt(self):
        """(mismatches, first bad bit pattern) of the distance kernels' short square root against the provable form over every
        float in [2^-96, inf) (include/mkamd_distance.h): (0, 0) on gfx950."""
        n, first = ctypes.c_uint64(0), ctypes.c_uint32(0)
        _check(load().mkamd_selftest_sqrt(self._h, ctypes.byref(n), ctypes.byref(first)))
        return int(n.value), int(first.value)

    def set_dist_kernels(self, avoid_mask: int = 0):
        """Kernels dist_trajectory must NOT take (include/mkamd_distance.h): 1 block-per-frame, 2 rows, 4 rectangular tiles, 8 the
        row kernel's 16-byte stores; 16: the row kernel wherever it applies; 32: host calls upload the whole coordinate array (no packing
        of the selected atoms' rows); 64: selfdist calls keep the pair-table kernel (no triangular row kernel); 128: short-row calls of few frames keep the tile kernel (no swapped row kernel); 0 = free choice.  Same bits whichever runs (tests, A-B timing)."""
        _check(load().mkamd_ctx_set_dist_kernels(self._h, int(avoid_mask)))

    def set_reduction_block(self, block: int = 0):
        """0: choose; 4 / 8: first-group atoms a wave of the closest-atom group reduction keeps in registers; -1: the generic kernel."""
        _check(load().mkamd_ctx_set_reduction_block(self._h, int(block)))

    def last_dist_kernel(self) -> str:
        """Kernels the last dist_trajectory call launched, as rocprofv3 prints them ('' before the first call)."""
        buf = ctypes.create_string_buffer(128)
        _check(load().mkamd_ctx_last_dist_kernel(self._h, buf, 128))
        return buf.value.decode()

    def pipelined_calls(self) -> int:
        """Lattice calls of this context whose pre-pass ran beside a previous call's tile kernel so far."""
        n = _c_i64(0)
        _check(load().mkamd_ctx_pipelined_calls(self._h, ctypes.byref(n)))
        return int(n.value)

    def enable_kernel_timing(self, on=True):
        _check(load().mkamd_ctx_enable_kernel_timing(self._h, int(bool(on))))

    def read_kernel_timing(self):
        ms, n = _c_dbl(0.0), _c_i64(0)
        _check(load().mkamd_ctx_read_kernel_timing(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        arch = ctypes.create_string_buffer(256)
        cus, mem = _c_int(0), ctypes.c_uint64(0)
        _check(load().mkamd_ctx_device_info(self._h, name, 256, ctypes.byref(cus), ctypes.byref(mem), arch, 256))
        return dict(name=name.value.decode(), arch=arch.value.decode(), compute_units=cus.value,
                    hbm_bytes=int(mem.value))

    # -- compute (raw pointers; see moleculekit_amd.batch for the numpy/torch wrappers) ---------
    def calculate_occupancy(self, centers, coords, sigmas, results):
        V, N, C = centers.shape[0], coords.shape[0], sigmas.shape[1]
        _check(load().mkamd_calculate_occupancy(self._h, _ptr(centers), V, _ptr(coords), N, _ptr(sigmas), C,
                                                _ptr(results)))

    def occupancy_centers_host(self, centers, coords, sigmas, sig_f64, C, box, out):
        _check(load().mkamd_occupancy_centers_host(self._h, _ptr(centers), centers.shape[0], _ptr(coords),
                                                   coords.shape[0], _ptr(sigmas), int(sig_f64), C, _ptr(box),
                                                   _ptr(out)))

    def occupancy_centers_dev(self, d_centers, V, d_coords, N, d_sigmas, sig_f64, C, box_host, d_out):
        _check(load().mkamd_occupancy_centers_dev(self._h, _ptr(d_centers), V, _ptr(d_coords), N, _ptr(d_sigmas),
                                                  int(sig_f64), C, _ptr(box_host), _ptr(d_out)))

    def voxelize_lattice_host(self, B, coords, offsets, sigmas, sig_f64, C, origins, nvox, voxelsize, box,
                              max_images, out):
        fn = load().mkamd_voxelize_lattice_host_f64 if out.dtype == np.float64 else load().mkamd_voxelize_lattice_host
        _check(fn(self._h, B, _ptr(coords), _ptr(offsets), _ptr(sigmas),
                                                  int(sig_f64), C, _ptr(origins), _ptr(nvox), float(voxelsize),
                                                  _ptr(box), int(max_images), _ptr(out)))

    def voxelize_lattice_host_begin(self, B, coords, offsets, sigmas, sig_f64, C, origins, nvox, voxelsize, box, max_images):
        """First half of voxelize_lattice_host: inputs shipped, kernels enqueued (the arrays must outlive ..._end)."""
        _check(load().mkamd_voxelize_lattice_host_begin(self._h, B, _ptr(coords), _ptr(offsets), _ptr(sigmas), int(sig_f64), C,
                                                        _ptr(origins), _ptr(nvox), float(voxelsize), _ptr(box), int(max_images)))
        self._begun = getattr(self, "_begun", 0) + 1          # which `begin` is pending (batch.voxelize_lattice_begin's finalizer)
        return self._begun

    def voxelize_lattice_host_end(self, out):
        """Second half: wait, result into `out` (float32 or float64, C-contiguous, B*V*C elements -- checked here and, the
        element count, by the library against what the pending call produced: it writes through a bare pointer)."""
        if not isinstance(out, np.ndarray) or out.dtype not in (np.float32, np.float64) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 or float64 ndarray")
        if out.dtype == np.float64:
            _check(load().mkamd_voxelize_lattice_host_end(self._h, None, _ptr(out), int(out.size)))
        else:
            _check(load().mkamd_voxelize_lattice_host_end(self._h, _ptr(out), None, int(out.size)))

    def voxelize_lattice_dev(self, B, d_coords, d_offsets, total_atoms, d_sigmas, sig_f64, C, d_origins, nvox,
                             voxelsize, d_box, max_images, d_out, d_affine=None):
        _check(load().mkamd_voxelize_lattice_aug_dev(self._h, B, _ptr(d_coords), _ptr(d_offsets), int(total_atoms),
                                                     _ptr(d_sigmas), int(sig_f64), C, _ptr(d_origins), _ptr(nvox),
                                                     float(voxelsize), _ptr(d_box), int(max_images), _ptr(d_affine),
                                                     _ptr(d_out)))

    def voxelize_lattice_topo_dev(self, B, d_coords, d_offsets, total_atoms, topology, d_origins, nvox, voxelsize, d_box, max_images, d_out,
                                  d_affine=None):
        """``voxelize_lattice_dev`` for items that are each one set of coordinates of ``topology``'s molecule (include/mkamd_voxel.h)."""
        _check(load().mkamd_voxelize_lattice_topo_dev(self._h, B, _ptr(d_coords), _ptr(d_offsets), int(total_atoms), topology._h,
                                                      _ptr(d_origins), _ptr(nvox), float(voxelsize), _ptr(d_box), int(max_images),
                                                      _ptr(d_affine), _ptr(d_out)))

    def grid_centers_host(self, bb_min, nvox, voxelsize, out):
        _check(load().mkamd_grid_centers_host(self._h, _ptr(bb_min), _ptr(nvox), float(voxelsize), _ptr(out)))

    # -- distance_utils row (include/mkamd_distance.h) -------------------------------------------------
    def dist_trajectory_host(self, coords, box, sel1, sel2, chains, selfdist, pbc, squared, out):
        N, _, F = coords.shape
        _check(load().mkamd_dist_trajectory_host(self._h, _ptr(coords), N, F, _ptr(box), _ptr(sel1), sel1.shape[0], _ptr(sel2),
                                                 sel2.shape[0], _ptr(chains), int(selfdist), int(pbc), int(squared), _ptr(out)))

    def dist_trajectory_dev(self, d_coords, F, d_box, d_sel1, n1, d_sel2, n2, d_chains, selfdist, pbc, squared, d_out):
        _check(load().mkamd_dist_trajectory_dev(self._h, _ptr(d_coords), F, _ptr(d_box), _ptr(d_sel1), n1, _ptr(d_sel2), n2,
                                                _ptr(d_chains), int(selfdist), int(pbc), int(squared), _ptr(d_out)))

    def contacts_trajectory_host(self, coords, box, sel1, sel2, chains, selfdist, pbc, threshold):
        """-> (frame_offsets int64 [F+1], pairs uint32 [n_contacts, 2]) -- thresholded and compacted on the GPU."""
        N, _, F = coords.shape
        offs = np.zeros(F + 1, dtype=np.int64)
        ptr = _vp(None)
        _check(load().mkamd_contacts_trajectory_host(self._h, _ptr(coords), N, F, _ptr(box), _ptr(sel1), sel1.shape[0], _ptr(sel2),
                                                     sel2.shape[0], _ptr(chains), int(selfdist), int(pbc), float(threshold),
                                                     _ptr(offs), ctypes.byref(ptr)))
        n = int(offs[-1])
        if n == 0 or not ptr.value:
            return offs, np.zeros((0, 2), dtype=np.uint32)
        buf = (ctypes.c_uint32 * (2 * n)).from_address(ptr.value)          # context-owned: copy before the next call
        return offs, np.frombuffer(buf, dtype=np.uint32).reshape(n, 2).copy()

    def dist_reduction_host(self, coords, box, g1a, g1o, g2a, g2o, ch1, ch2, selfdist, pairs, pbc, masses, r1, r2, out):
        N, _, F = coords.shape
        _check(load().mkamd_dist_reduction_host(self._h, _ptr(coords), N, F, _ptr(box), _ptr(g1a), _ptr(g1o), g1o.shape[0] - 1,
                                                _ptr(g2a), _ptr(g2o), g2o.shape[0] - 1, _ptr(ch1), _ptr(ch2), int(selfdist),
                                                int(pairs), int(pbc), _ptr(masses), int(r1), int(r2), _ptr(out)))

    # device-resident forms: arguments are torch CUDA tensors / objects with data_ptr() (or raw addresses); asynchronous on the
    # context's stream except the contact list
    def contacts_trajectory_dev(self, d_coords, F, d_box, d_sel1, n1, d_sel2, n2, d_chains, selfdist, pbc, threshold):
        """-> (frame_offsets int64 [F+1] host, address of the device list of 2 * n uint32 (context-owned until the next contacts
        call; 0 when empty), n)."""
        offs = np.zeros(F + 1, dtype=np.int64)
        ptr = _vp(None)
        _check(load().mkamd_contacts_trajectory_dev(self._h, _ptr(d_coords), F, _ptr(d_box), _ptr(d_sel1), n1, _ptr(d_sel2), n2, _ptr(d_chains),
                                                    int(selfdist), int(pbc), float(threshold), _ptr(offs), ctypes.byref(ptr)))
        return offs, int(ptr.value or 0), int(offs[-1])

    def dist_reduction_dev(self, d_coords, N, F, d_box, d_g1a, d_g1o, ng1, n_g1_atoms, d_g2a, d_g2o, ng2, d_ch1, d_ch2, selfdist, pairs, pbc,
                           d_masses, r1, r2, d_out):
        _check(load().mkamd_dist_reduction_dev(self._h, _ptr(d_coords), N, F, _ptr(d_box), _ptr(d_g1a), _ptr(d_g1o), ng1, n_g1_atoms, _ptr(d_g2a),
                                               _ptr(d_g2o), ng2, _ptr(d_ch1), _ptr(d_ch2), int(selfdist), int(pairs), int(pbc), _ptr(d_masses),
                                               int(r1), int(r2), _ptr(d_out)))

    def cdist_dev(self, d_c1, n1, d_c2, n2, dim, d_out):
        _check(load().mkamd_cdist_dev(self._h, _ptr(d_c1), n1, _ptr(d_c2), n2, dim, _ptr(d_out)))

    def pdist_dev(self, d_c, n, dim, d_out):
        _check(load().mkamd_pdist_dev(self._h, _ptr(d_c), n, dim, _ptr(d_out)))

    def cdist_host(self, c1, c2, out):
        _check(load().mkamd_cdist_host(self._h, _ptr(c1), c1.shape[0], _ptr(c2), c2.shape[0], c1.shape[1], _ptr(out)))

    def pdist_host(self, c, out):
        _check(load().mkamd_pdist_host(self._h, _ptr(c), c.shape[0], c.shape[1], _ptr(out)))

    def grid_centers_dev(self, bb_min, nvox, voxelsize, d_out):
        _check(load().mkamd_grid_centers_dev(self._h, _ptr(bb_min), _ptr(nvox), float(voxelsize), _ptr(d_out)))


class Topology:
    """What the voxelizer's pre-pass derives from a molecule's sigmas ALONE, kept on the device for every later call over
    coordinates of that molecule (include/mkamd_voxel.h, (3c)): ``sigmas`` is the molecule's [n_atoms, C] matrix -- a numpy
    array (float32 / float64) or a CUDA tensor of the context's device.  The library keeps its own copy."""

    def __init__(self, ctx, sigmas, voxelsize):
        self._h = _vp(None)
        self._ctx = ctx
        h = _vp(None)
        if isinstance(sigmas, np.ndarray) or not hasattr(sigmas, "data_ptr"):
            sig = np.ascontiguousarray(sigmas)
            if sig.dtype not in (np.float32, np.float64):
                sig = sig.astype(np.float64)
            if sig.ndim != 2:
                raise ValueError("sigmas must be (natoms, nchannels)")
            _check(load().mkamd_topology_create_host(ctx._h, _ptr(sig), int(sig.dtype == np.float64), int(sig.shape[0]), int(sig.shape[1]),
                                                     float(voxelsize), ctypes.byref(h)))
            self.n_atoms, self.n_channels = int(sig.shape[0]), int(sig.shape[1])
        else:
            import torch
            if not (sigmas.is_cuda and sigmas.dim() == 2 and sigmas.is_contiguous() and sigmas.dtype in (torch.float32, torch.float64)):
                raise ValueError("sigmas must be a contiguous float32 / float64 CUDA tensor (natoms, nchannels)")
            ctx.set_stream(torch.cuda.current_stream(sigmas.device).cuda_stream)
            _check(load().mkamd_topology_create_dev(ctx._h, int(sigmas.data_ptr()), int(sigmas.dtype == torch.float64), int(sigmas.shape[0]),
                                                    int(sigmas.shape[1]), float(voxelsize), ctypes.byref(h)))
            self.n_atoms, self.n_channels = int(sigmas.shape[0]), int(sigmas.shape[1])
        self._h = h
        self.voxelsize = float(voxelsize)

    @property
    def has_wide_sigmas(self) -> bool:
        w = _c_i32(0)
        _check(load().mkamd_topology_info(self._h, None, None, None, ctypes.byref(w)))
        return bool(w.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            ctx = getattr(self, "_ctx", None)
            alive = ctx is not None and getattr(ctx, "_h", None) is not None and ctx._h.value
            load().mkamd_topology_destroy(ctx._h if alive else None, self._h)
            self._h = _vp(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context(device: int | None = None) -> Context:
    """The calling THREAD's context on ``device`` (``None`` -> MKAMD_DEVICE / LOCAL_RANK env or 0).

    A context owns one stream, one grow-only workspace and pinned staging buffers, and ctypes releases the GIL
    during every call, so two host threads must never drive the same context at once (include/mkamd_voxel.h: "one
    context per device and per host thread").  The reference's Cython kernel runs under the GIL and is thread-safe
    by construction; a default context per (process, thread, device) gives the drop-in API the same guarantee.

    The contexts live in the thread's own ``threading.local`` storage: nobody but the owning thread ever looks one up,
    and nothing closes it from outside -- it is released (``Context.__del__``) when the thread's storage dies AND no
    one else holds a reference (a context handed to another object, e.g. ``ShardedVoxelizer``, outlives its thread).
    A thread that Python did not start (a C++ callback thread) gets storage like any other.  After ``fork`` the child
    makes its own contexts and leaves the parent's handles alone."""
    if device is None:
        device = int(os.environ.get("MKAMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n > 0:
            device %= n
    held = getattr(_tls, "contexts", None)
    if held is None or getattr(_tls, "pid", None) != os.getpid():
        if held:                                   # forked child: the parent's device handles are not ours to free
            for c in held.values():
                c._h = _vp(None)
        held = _tls.contexts = {}
        _tls.pid = os.getpid()
    ctx = held.get(int(device))
    if ctx is None:
        ctx = held[int(device)] = Context(device)
    return ctx
