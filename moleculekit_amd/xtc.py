"""XTC trajectory reading on top of libmkamd.so's host-side decoder.

Mirrors the reference's low-level module ``moleculekit.xtc`` (fileformats/xtc/xtc.pyx: ``get_xtc_natoms``,
``get_xtc_nframes``, ``read_xtc`` :34-53, ``read_xtc_frames`` :57-81 -- same arrays, same float32 bits, units as in the
file: nm / ps) and ``readers.XTCread`` (readers.py:1830-1866: the conversion to Angstrom / fs and from box vectors to
lengths and angles, unitcell.py:128-221).  Frames are decoded in parallel on host threads.  ``XTCread`` returns a
plain ``Trajectory`` record (coords float32 [N,3,F] in A, box [3,F], boxangles, step, time) -- the arrays
``batch.iterVoxelizeTrajectory`` / ``voxelizeTrajectory`` take; building a ``Molecule`` around it is outside this package.
"""
from __future__ import annotations

import ctypes
from collections import namedtuple

import numpy as np

from . import _lib

Trajectory = namedtuple("Trajectory", ["coords", "box", "boxangles", "step", "time"])


def _path(filename) -> bytes:
    return filename if isinstance(filename, bytes) else str(filename).encode("UTF-8")


def _info(filename):
    na, nf = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib._check(_lib.load().mkamd_xtc_info(_path(filename), ctypes.byref(na), ctypes.byref(nf)))
    return na.value, nf.value


def get_xtc_natoms(filename) -> int:
    return _info(filename)[0]


def get_xtc_nframes(filename) -> int:
    return _info(filename)[1]


def _read(filename, frames, nthreads):
    natoms, nframes = _info(filename)
    if frames is None:
        sel, n = None, nframes
    else:
        sel = np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
        n = len(sel)
    coords = np.zeros((natoms, 3, n), dtype=np.float32)
    box = np.zeros((3, 3, n), dtype=np.float32)
    time = np.zeros(n, dtype=np.float32)
    step = np.zeros(n, dtype=np.int32)
    _lib._check(_lib.load().mkamd_xtc_read(_path(filename), _lib._ptr(sel), n, natoms, _lib._ptr(coords), _lib._ptr(box),
                                           _lib._ptr(time), _lib._ptr(step), int(nthreads)))
    return coords, box, time, step


def read_xtc(filename, nthreads: int = 0):
    """All frames: ``(coords f32 [N,3,F] nm, boxvectors f32 [3,3,F] nm, time f32 [F] ps, step i32 [F])``."""
    return _read(filename, None, nthreads)


def read_xtc_frames(filename, frames, nthreads: int = 0):
    """The listed frames (any order), same return layout as ``read_xtc``."""
    return _read(filename, frames, nthreads)


def box_vectors_to_lengths_and_angles(a, b, c):
    """Box lengths and angles (degrees) from box vectors, per frame (unitcell.py:128-221); all-zero vectors give
    zero lengths and 90 degree angles."""
    a, b, c = (np.asarray(v) for v in (a, b, c))
    if np.all(a == 0) or np.all(b == 0) or np.all(c == 0):
        nf = a.shape[:-1]
        return np.zeros(nf), np.zeros(nf), np.zeros(nf), np.full(nf, 90.0), np.full(nf, 90.0), np.full(nf, 90.0)
    la, lb, lc = (np.sqrt(np.sum(v * v, axis=-1)) for v in (a, b, c))
    ang = lambda u, v, lu, lv: np.arccos(np.einsum("...i, ...i", u, v) / (lu * lv)) * 180.0 / np.pi
    return la, lb, lc, ang(b, c, lb, lc), ang(c, a, lc, la), ang(a, b, la, lb)


def XTCread(filename, frame=None, nthreads: int = 0) -> Trajectory:
    """``readers.XTCread`` (readers.py:1830-1866): coordinates and box in Angstrom (x10), time in fs (x1e3); step
    and time replaced by 0..F-1 / zeros when the file holds none (all-zero sums), like the reference."""
    if frame is None:
        coords, boxvectors, time, step = read_xtc(filename, nthreads)
    else:
        coords, boxvectors, time, step = read_xtc_frames(filename, np.atleast_1d(frame), nthreads)
    if coords.shape[2] == 0:
        raise RuntimeError(f"Malformed XTC file. No frames read from: {filename}")
    if coords.shape[0] == 0:
        raise RuntimeError(f"Malformed XTC file. No atoms read from: {filename}")
    time = time.astype(np.float64)
    coords *= 10.0
    boxvectors *= 10.0
    time *= 1e3
    nframes = coords.shape[2]
    if np.sum(step) == 0:
        step = np.arange(nframes, dtype=np.uint64)
    if np.sum(time) == 0:
        time = np.zeros(nframes, dtype=np.float64)
    bx, by, bz, alpha, beta, gamma = box_vectors_to_lengths_and_angles(boxvectors[0].T, boxvectors[1].T, boxvectors[2].T)
    return Trajectory(coords=coords, box=np.stack([bx, by, bz], axis=0), boxangles=np.stack([alpha, beta, gamma], axis=0),
                      step=step, time=time)


# ------------------------------------------------------------------------------------------------
# decoding on the device (round 4; include/mkamd_xtc.h, csrc/xtc_gpu.h): a GPU lane decodes a frame
# ------------------------------------------------------------------------------------------------
DESC_BYTES = 64        # sizeof(mkamd::XtcFrameDesc)
XTC_PAD = 1024         # bytes the device copy of a chunk's records extends past them (MKAMD_XTC_PAD, include/mkamd_xtc.h)


DESC_DTYPE = np.dtype([("data_off", "<u8"), ("nbytes", "<u4"), ("smallidx", "<i4"), ("lo", "<i4", 3), ("range", "<u4", 3),
                       ("inv_precision", "<f4"), ("triple_bits", "<i4"), ("field_bits", "<i4", 3), ("raw", "<i4")])
assert DESC_DTYPE.itemsize == DESC_BYTES


def device_decodable(desc, natoms) -> bool:
    """What the headers tell about ``status 2`` of the device decoder (include/mkamd_xtc.h): no frame of ``desc`` (from
    ``chunk_desc``) packs a coordinate into more than 64 bits, has >= 2^21 atoms or a stream of >= 512 MB.  (A run of small
    atoms coded in more than 64 bits is only seen by the device; files with such runs have ranges that fail here.)"""
    d = np.ascontiguousarray(desc).view(DESC_DTYPE).reshape(-1)
    return bool(natoms < (1 << 21) and np.all(d["triple_bits"] <= 64) and np.all(d["nbytes"] < (1 << 29) - 4))


def chunk_desc(filename, frames, natoms):
    """Host half of the device decoder for the frames ``frames`` (int64 array): ``(desc uint8 [n, 64], byte_lo, byte_hi,
    boxvectors f32 [3,3,n] nm, time f32 [n] ps, step i32 [n])`` -- record headers only, no coordinate is decoded."""
    sel = np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
    n = len(sel)
    desc = np.empty((n, DESC_BYTES), dtype=np.uint8)
    box = np.empty((3, 3, n), dtype=np.float32)
    time = np.empty(n, dtype=np.float32)
    step = np.empty(n, dtype=np.int32)
    lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib._check(_lib.load().mkamd_xtc_chunk_desc(_path(filename), _lib._ptr(sel), n, int(natoms), _lib._ptr(desc), ctypes.byref(lo),
                                                 ctypes.byref(hi), _lib._ptr(box), _lib._ptr(time), _lib._ptr(step)))
    return desc, int(lo.value), int(hi.value), box, time, step


def byte_range(filename, frames, natoms):
    """``(byte_lo, byte_hi)`` of the records of ``frames`` from the frame index alone (no record is read)."""
    sel = np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
    lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib._check(_lib.load().mkamd_xtc_byte_range(_path(filename), _lib._ptr(sel), len(sel), int(natoms), ctypes.byref(lo), ctypes.byref(hi)))
    return int(lo.value), int(hi.value)


def chunk_desc_mem(filename, frames, natoms, bytes_ptr, bytes_lo, bytes_hi):
    """``chunk_desc`` from a host COPY of the file's bytes ``[bytes_lo, bytes_hi)`` at address ``bytes_ptr`` (the pinned staging buffer
    ``mkamd_xtc_copy_bytes`` has just filled): the same tuple, without touching the file's pages once more."""
    sel = np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
    n = len(sel)
    desc = np.empty((n, DESC_BYTES), dtype=np.uint8)
    box = np.empty((3, 3, n), dtype=np.float32)
    time = np.empty(n, dtype=np.float32)
    step = np.empty(n, dtype=np.int32)
    lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib._check(_lib.load().mkamd_xtc_chunk_desc_mem(_path(filename), _lib._ptr(sel), n, int(natoms), ctypes.c_void_p(int(bytes_ptr)), int(bytes_lo),
                                                     int(bytes_hi), _lib._ptr(desc), ctypes.byref(lo), ctypes.byref(hi), _lib._ptr(box), _lib._ptr(time),
                                                     _lib._ptr(step)))
    return desc, int(lo.value), int(hi.value), box, time, step


def read_xtc_frames_dev(filename, frames=None, scale: float = 1.0, ctx=None):
    """``read_xtc_frames`` with the coordinates decoded ON THE GPU: returns ``(xyz, boxvectors, time, step)`` with ``xyz`` a
    float32 CUDA tensor ``[n, natoms, 3]`` (frame-major: the voxelizer's packed items; ``scale`` 10 gives Angstrom) whose
    values are bit for bit ``read_xtc_frames(...)[0] * scale`` transposed.  Synchronous (tests, one-off reads); the streaming
    form is ``batch.iterVoxelizeXTC`` (``decode="auto"`` / ``"gpu"``).  Raises for a stream the device decoder refuses (ranges
    whose mixed-radix number exceeds 64 bits, >= 2^21 atoms: use the host decoder)."""
    import torch

    ctx = ctx or _lib.default_context()
    natoms, nframes = _info(filename)
    sel = np.arange(nframes, dtype=np.int64) if frames is None else np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
    dev = torch.device("cuda", ctx.device)
    n = len(sel)
    xyz = torch.empty((n, natoms, 3), dtype=torch.float32, device=dev)
    if n == 0:
        return xyz, np.zeros((3, 3, 0), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32)
    desc, lo, hi, box, time, step = chunk_desc(filename, sel, natoms)
    raw = torch.zeros(hi - lo + XTC_PAD, dtype=torch.uint8).pin_memory()          # (the kernel reads ahead of what it uses)
    _lib._check(_lib.load().mkamd_xtc_copy_bytes(_path(filename), lo, hi, raw.data_ptr(), 0))
    with torch.cuda.device(dev):
        d_raw = raw.to(dev, non_blocking=True)
        d_desc = torch.as_tensor(desc, device=dev)
        d_st = torch.empty(n, dtype=torch.int32, device=dev)
        work = torch.empty(int(_lib.load().mkamd_xtc_decode_work_bytes(n, natoms)), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream(dev)
        _lib._check(_lib.load().mkamd_xtc_decode_dev(ctx._h, s.cuda_stream or None, d_raw.data_ptr(), d_desc.data_ptr(), n, natoms,
                                                     float(scale), xyz.data_ptr(), d_st.data_ptr(), work.data_ptr(), work.numel()))
        st = d_st.cpu().numpy()
    if st.any():
        bad = int(np.flatnonzero(st)[0])
        raise RuntimeError(f"device XTC decoder: frame {int(sel[bad])} " + ("is corrupt" if st[bad] == 1 else
                           "is outside what the device decoder takes (numbers of more than 64 bits -- a coordinate range beyond ~2 million "
                           "quanta per axis --, >= 2^21 atoms or >= 512 MB per frame): use the host decoder"))
    return xyz, box, time, step


# ------------------------------------------------------------------------------------------------
# writing (round 4): enough of an encoder to produce valid trajectories for tests and benchmarks
# ------------------------------------------------------------------------------------------------
_MAGIC = 1995
_FIRSTIDX = 9          # index of the first usable entry of the format's table of "magic" sizes (xdrfile.cpp:545)


def _pack_bits(*fields):
    """Per atom the bit fields ``(values uint64 [N], nbits)`` one after the other, most significant bit first; the atoms
    concatenated -> uint8 bytes, zero-padded to a whole byte."""
    cols = []
    for values, nbits in fields:
        shifts = np.arange(nbits - 1, -1, -1, dtype=np.uint64)
        cols.append(((values[:, None] >> shifts[None, :]) & np.uint64(1)).astype(np.uint8))
    return np.packbits(np.concatenate(cols, axis=1).reshape(-1))


def write_xtc(filename, coords, box, time, step, precision: float = 1000.0):
    """``moleculekit.xtc.write_xtc`` (fileformats/xtc/xtc.pyx:86-97): coords float32 [N,3,F] in nm, box vectors float32
    [3,3,F] in nm, time float32 [F] (ps), step [F].  Appends nothing, overwrites ``filename``.

    The coordinate block is the format's compressed form (xdrfile.cpp: xdrfile_compress_coord_float) in its simplest legal
    shape: every atom is written as one "large" coordinate -- the three integers ``round(x * precision) - min`` folded into
    one mixed-radix number of ``bits(prod(range))`` bits, least-significant byte first -- followed by a 0 flag bit ("run
    length unchanged": it starts at 0, so no atom is ever coded as a small difference to its predecessor).  The
    reference's writer finds runs of near neighbours (water molecules) and codes them in fewer bits; a decoder reads both
    alike, and this one costs ~5 bytes per atom of a 67 A box instead of ~4.  Up to nine atoms are stored as plain floats,
    like the reference does."""
    import struct

    coords = np.ascontiguousarray(coords, dtype=np.float32)
    if coords.ndim != 3 or coords.shape[1] != 3:
        raise ValueError("coords must be (natoms, 3, nframes)")
    N, _, F = coords.shape
    box = np.ascontiguousarray(box, dtype=np.float32).reshape(3, 3, -1)
    if box.shape[2] != F:
        raise ValueError("box must hold one set of box vectors per frame: (3, 3, nframes)")
    time = np.ascontiguousarray(time, dtype=np.float32).reshape(-1)
    step = np.ascontiguousarray(step).astype(np.int64).reshape(-1)
    if len(time) != F or len(step) != F:
        raise ValueError("time and step must have one entry per frame")
    prec = np.float32(precision)
    with open(_path(filename), "wb") as fh:
        for f in range(F):
            fh.write(struct.pack(">iii", _MAGIC, N, int(step[f]) & 0x7fffffff))
            fh.write(struct.pack(">f", float(time[f])))
            fh.write(box[:, :, f].astype(">f4").tobytes())               # row-major box vectors
            fh.write(struct.pack(">i", N))
            x = coords[:, :, f]
            if N <= 9:
                fh.write(x.astype(">f4").tobytes())
                continue
            # xdrfile.cpp:606-624: lf = x * precision; (int)(lf + 0.5) for lf >= 0, (int)(lf - 0.5) below (float arithmetic)
            lf = x * prec
            if not np.all(np.abs(lf) < 2.0e9):
                raise ValueError("coordinate too large for the XTC integer range at this precision")
            ints = np.where(lf >= 0, lf + np.float32(0.5), lf - np.float32(0.5)).astype(np.int64)
            lo, hi = ints.min(0), ints.max(0)
            size = (hi - lo + 1).astype(np.int64)
            rel = (ints - lo).astype(np.uint64)
            head = struct.pack(">f", float(prec)) + struct.pack(">iii", *[int(v) for v in lo]) + struct.pack(">iii", *[int(v) for v in hi])
            if np.any(size > 0xffffff):
                # per-axis fields (xdrfile.cpp:634-640: bitsize = 0)
                nb = [int(int(s).bit_length()) for s in size]
                payload = _pack_bits((rel[:, 0], nb[0]), (rel[:, 1], nb[1]), (rel[:, 2], nb[2]), (np.zeros(N, np.uint64), 1))
            else:
                prod = int(size[0]) * int(size[1]) * int(size[2])
                nbits = prod.bit_length()                                # sizeofints(3, sizeint), xdrfile.cpp:425-457
                if nbits <= 63:
                    V = (rel[:, 0] * np.uint64(size[1]) + rel[:, 1]) * np.uint64(size[2]) + rel[:, 2]
                else:                                                    # three ranges of up to 24 bits: up to 72 bits -- Python integers
                    V = (rel[:, 0].astype(object) * int(size[1]) + rel[:, 1].astype(object)) * int(size[2]) + rel[:, 2].astype(object)
                # the number goes out least-significant BYTE first, its top (nbits mod 8, or 8) bits last (sendints, :489-543)
                nfull, top = (nbits - 1) // 8, nbits - 8 * ((nbits - 1) // 8)
                if nbits <= 63:
                    fields = [((V >> np.uint64(8 * b)) & np.uint64(0xff), 8) for b in range(nfull)]
                    fields.append(((V >> np.uint64(8 * nfull)) & np.uint64((1 << top) - 1), top))
                else:
                    byte = lambda sh, mask: np.array([(int(v) >> sh) & mask for v in V], dtype=np.uint64)
                    fields = [(byte(8 * b, 0xff), 8) for b in range(nfull)]
                    fields.append((byte(8 * nfull, (1 << top) - 1), top))
                fields.append((np.zeros(N, np.uint64), 1))               # ... then the flag bit: 0
                payload = _pack_bits(*fields)
            fh.write(head + struct.pack(">i", _FIRSTIDX))                # smallidx: any valid index (never used: no small atoms)
            fh.write(struct.pack(">i", len(payload)))
            fh.write(payload.tobytes() + b"\0" * ((4 - len(payload) % 4) % 4))
