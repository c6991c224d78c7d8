"""Batched / device-resident entry points of the voxel-descriptor hot path.

The reference voxelizes ONE molecule per Python call (tools/voxeldescriptors.py:251-365).  On an
MI355X a single 24^3 grid is far too small to fill 256 CUs, so the native unit of work here is a
BATCH of independent items (ligand poses, molecules of a virtual screen, trajectory frames) packed
back to back and voxelized by one launch sequence of ``libmkamd.so``.

Two flavours:
  * numpy in / numpy out           -> ``voxelize_lattice``        (H2D + kernels + D2H)
  * torch CUDA tensors in / out    -> ``voxelize_lattice_torch``  (inputs and features stay in HBM;
                                      this is what bench.py and ML data loaders use)
torch is plumbing only here: it owns device memory and the stream; all compute is in the HIP library.
"""
from __future__ import annotations

import numpy as np

from . import _lib

CUTOFF = 5.0  # A; occupancy_utils.pyx:53


def pack_items(coords_list, sigmas_list):
    """Pack per-item arrays back to back -> (coords f32 [sumN,3], sigmas [sumN,C], offsets i64 [B+1])."""
    ns = [int(np.asarray(c).shape[0]) for c in coords_list]
    offs = np.zeros(len(ns) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(ns)
    if len(ns) == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float64), offs
    coords = np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1, 3) for c in coords_list])
    sig = np.concatenate([np.asarray(s) for s in sigmas_list])
    return coords, sig, offs


def max_images_per_atom(box, nvoxels, voxelsize) -> int:
    """Upper bound on the periodic images of one atom inside grid + cutoff halo (include/mkamd_voxel.h)."""
    box = np.asarray(box, dtype=np.float64).reshape(-1, 3)
    if box.size == 0:
        return 1
    if not np.all(box > 2 * CUTOFF):
        raise ValueError("periodic box edges must be > 10 A (2 x cutoff)")
    span = (np.maximum(np.asarray(nvoxels, dtype=np.float64) - 1, 0)) * voxelsize + 2 * CUTOFF + 2e-3 * voxelsize
    m = np.prod(np.floor(span[None, :] / box) + 1, axis=1)
    return int(m.max())


def _sigma_array(sigmas):
    sigmas = np.asarray(sigmas)
    if sigmas.dtype == np.float32:
        return np.ascontiguousarray(sigmas), 0
    return np.ascontiguousarray(sigmas, dtype=np.float64), 1


def voxelize_lattice(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None, out=None, ctx=None):
    """Voxelize B packed items onto identical lattice grids (host arrays in, float32 out).

    coords f32 [sumN,3]; atom_offsets i64 [B+1]; sigmas f64|f32 [sumN,C]; origins f64 [B,3] =
    position of voxel (0,0,0) of each item (``bb_min`` of getCenters, voxeldescriptors.py:234-243);
    nvoxels (3,); box f32 [B,3] or None (orthorhombic minimum image, distance_utils.pyx:49-52).
    Returns features float32 [B, V, C], V flattened x slowest / z fastest (reference layout).
    """
    ctx = ctx or _lib.default_context()
    coords = np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 3)
    atom_offsets = np.ascontiguousarray(atom_offsets, dtype=np.int64)
    sigmas, sig64 = _sigma_array(sigmas)
    if sigmas.ndim != 2 or sigmas.shape[0] != coords.shape[0]:
        raise ValueError("sigmas must be (natoms, nchannels) matching coords")
    origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    B = origins.shape[0]
    if atom_offsets.shape != (B + 1,) or (B >= 0 and atom_offsets[-1] != coords.shape[0]):
        raise ValueError("atom_offsets must have B+1 entries ending at the total atom count")
    nv = np.ascontiguousarray(nvoxels, dtype=np.int32).reshape(3)
    C = int(sigmas.shape[1])
    V = int(np.prod(nv.astype(np.int64)))
    bx = None
    if box is not None:
        bx = np.ascontiguousarray(box, dtype=np.float32).reshape(B, 3)
    if out is None:
        out = np.empty((B, V, C), dtype=np.float32)
    elif out.dtype not in (np.float32, np.float64) or not out.flags["C_CONTIGUOUS"] or out.size != B * V * C:
        raise ValueError("out must be a C-contiguous float32 (or float64: widened by the library) array of B*V*C elements")
    ctx.voxelize_lattice_host(B, coords, atom_offsets, sigmas, sig64, C, origins, nv, float(voxelsize), bx, 0, out)
    return out.reshape(B, V, C)


def voxelize_lattice_begin(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None, ctx=None, dtype=np.float32):
    """``voxelize_lattice`` in two halves: the inputs are shipped and the kernels enqueued here; the function returned
    waits and hands back features [B, V, C] of ``dtype`` (float32, or float64: widened by the library).  Host work done
    between the two runs beside the device (the drop-in getVoxelDescriptors copies its voxel centres there)."""
    ctx = ctx or _lib.default_context()
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("dtype must be float32 or float64 (the library writes 4- or 8-byte floats into the result)")
    coords = np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 3)
    atom_offsets = np.ascontiguousarray(atom_offsets, dtype=np.int64)
    sigmas, sig64 = _sigma_array(sigmas)
    if sigmas.ndim != 2 or sigmas.shape[0] != coords.shape[0]:
        raise ValueError("sigmas must be (natoms, nchannels) matching coords")
    origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    B = origins.shape[0]
    if atom_offsets.shape != (B + 1,) or (B >= 0 and atom_offsets[-1] != coords.shape[0]):
        raise ValueError("atom_offsets must have B+1 entries ending at the total atom count")
    nv = np.ascontiguousarray(nvoxels, dtype=np.int32).reshape(3)
    C = int(sigmas.shape[1])
    V = int(nv[0]) * int(nv[1]) * int(nv[2])
    bx = None if box is None else np.ascontiguousarray(box, dtype=np.float32).reshape(B, 3)
    ticket = ctx.voxelize_lattice_host_begin(B, coords, atom_offsets, sigmas, sig64, C, origins, nv, float(voxelsize), bx, 0)
    return _PendingHostCall(ctx, ticket, (B, V, C), dtype, (coords, atom_offsets, sigmas, origins, nv, bx))


def _abandon_host_call(ctx_ref, ticket):
    ctx = ctx_ref()
    # only the call this object began: a later `begin` (which abandons an unfinished one itself) is somebody else's
    if ctx is not None and getattr(ctx, "_begun", None) == ticket:
        try:
            ctx.abandon_pending()
        except Exception:
            pass


class _PendingHostCall:
    """The second half of ``voxelize_lattice_begin``: call it once for the features.  Dropped without having been called
    (an exception between the halves, a KeyboardInterrupt) it gives the pending call up, so that the context -- often the
    thread's shared default context -- takes every entry point again (include/mkamd_voxel.h, mkamd_ctx_abandon_pending)."""

    def __init__(self, ctx, ticket, shape, dtype, keep):
        import weakref
        self._ctx, self._shape, self._dtype, self._keep = ctx, shape, dtype, keep   # the inputs outlive the call
        self._finalizer = weakref.finalize(self, _abandon_host_call, weakref.ref(ctx), ticket)

    def __call__(self):
        self._finalizer.detach()                 # the library ends (or has already abandoned) the call itself from here on
        out = np.empty(self._shape, dtype=self._dtype)
        self._ctx.voxelize_lattice_host_end(out)
        return out

    def abandon(self):
        """Give the call up explicitly (what garbage collection does for a dropped object)."""
        if self._finalizer.alive:
            self._finalizer()


def occupancy_centers(centers, coords, sigmas, box=None, ctx=None):
    """Arbitrary (non-lattice) centres: float32 [V, C] occupancies (calculate_occupancy semantics)."""
    ctx = ctx or _lib.default_context()
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    sigmas, sig64 = _sigma_array(sigmas)
    if centers.ndim != 2 or centers.shape[1] != 3 or coords.ndim != 2 or coords.shape[1] != 3:
        raise ValueError("centers and coords must be (n, 3)")
    if sigmas.ndim != 2 or sigmas.shape[0] != coords.shape[0]:
        raise ValueError("sigmas must be (natoms, nchannels) matching coords")
    out = np.empty((centers.shape[0], sigmas.shape[1]), dtype=np.float32)
    bx = None if box is None else np.ascontiguousarray(box, dtype=np.float64).reshape(3)
    ctx.occupancy_centers_host(centers, coords, sigmas, sig64, int(sigmas.shape[1]), bx, out)
    return out


def grid_centers(bb_min, nvoxels, voxelsize, ctx=None):
    """float64 [V,3] lattice centres generated on the GPU, bit-exact with getCenters."""
    ctx = ctx or _lib.default_context()
    bb = np.ascontiguousarray(bb_min, dtype=np.float64).reshape(3)
    nv = np.ascontiguousarray(nvoxels, dtype=np.int32).reshape(3)
    out = np.empty((int(np.prod(nv.astype(np.int64))), 3), dtype=np.float64)
    ctx.grid_centers_host(bb, nv, float(voxelsize), out)
    return out


# ------------------------------------------------------------------------------------------------
# device-resident flavour (torch tensors are only containers for HBM pointers)
# ------------------------------------------------------------------------------------------------
def rotation_affines(rotations, centers):
    """float64 [B,12] rigid transforms equal to ``rotateCoordinates(coords, rotations[b], centers[b])``
    (voxeldescriptors.py:78-114: rotations about x, then y, then z around ``center``): row-major 3x3 matrix,
    then the translation."""
    from .util import rotationMatrix
    rotations = np.asarray(rotations, dtype=np.float64).reshape(-1, 3)
    centers = np.broadcast_to(np.asarray(centers, dtype=np.float64).reshape(-1, 3), rotations.shape)
    out = np.empty((rotations.shape[0], 12), dtype=np.float64)
    for b, (r, c) in enumerate(zip(rotations, centers)):
        m = rotationMatrix([0, 0, 1], r[2]) @ rotationMatrix([0, 1, 0], r[1]) @ rotationMatrix([1, 0, 0], r[0])
        out[b, :9] = m.ravel()
        out[b, 9:] = c - m @ c
    return out


def voxelize_lattice_torch(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None,
                           max_images=1, out=None, ctx=None, channel_first=False, affine=None, topology=None):
    """Same as ``voxelize_lattice`` on torch CUDA tensors; asynchronous on torch's current stream.

    ``topology`` (``_lib.Topology``, with ``sigmas=None``): every item is one set of coordinates of that molecule -- the frames
    of a trajectory; what the pre-pass derives from the sigmas is taken from the handle (include/mkamd_voxel.h (3c)): the same
    features bit for bit, fewer instructions per call.

    coords float32 [sumN,3], atom_offsets int64 [B+1], sigmas float32|float64 [sumN,C],
    origins float64 [B,3], box float32 [B,3] or None (pass ``max_images`` from
    ``max_images_per_atom`` when the box is smaller than grid + 10 A).
    ``affine``: optional float64 [B,12] CUDA tensor of per-item rigid transforms (``rotation_affines``) fused into
    the binning stage -- on-device augmentation, nothing leaves HBM.
    Returns float32 [B,V,C] on the same device, or with ``channel_first`` the logical [B,C,nx,ny,nz] tensor a
    ``nn.Conv3d`` takes -- as a zero-copy view: voxel-major / channel-minor storage is exactly PyTorch's
    ``torch.channels_last_3d`` (NDHWC) memory format.
    """
    import torch

    dev = coords.device
    if dev.type != "cuda":
        raise RuntimeError("voxelize_lattice_torch needs CUDA/HIP tensors (there is no CPU path)")
    dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
    if ctx is not None and ctx.device != dev_index:
        raise ValueError(f"ctx lives on GPU {ctx.device} but the tensors are on cuda:{dev_index} (kernels run on the context's device)")
    ctx = ctx or _lib.default_context(dev_index)
    if (sigmas is None) == (topology is None):
        ctx.withdraw_promise()
        raise ValueError("pass the sigmas or a topology (not both)")
    if not (coords.dtype == torch.float32 and coords.is_contiguous() and atom_offsets.dtype == torch.int64 and atom_offsets.is_contiguous()
            and (sigmas is None or (sigmas.dtype in (torch.float32, torch.float64) and sigmas.is_contiguous() and sigmas.dim() == 2))
            and origins.dtype == torch.float64 and origins.is_contiguous()):
        ctx.withdraw_promise()            # (see below)
        raise AssertionError("voxelize_lattice_torch: coords float32, atom_offsets int64, sigmas float32|float64 [n, C], origins float64, all contiguous")
    B = int(origins.shape[0])
    C = int(sigmas.shape[1]) if topology is None else topology.n_channels
    nv = np.ascontiguousarray(nvoxels, dtype=np.int32).reshape(3)
    V = int(np.prod(nv.astype(np.int64)))
    if out is None:
        out = torch.empty((B, V, C), dtype=torch.float32, device=dev)
    else:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == B * V * C
    d_box = None
    if box is not None:
        assert box.dtype == torch.float32 and box.is_contiguous()
        d_box = box.data_ptr()
    d_aff = None
    if affine is not None:
        assert affine.dtype == torch.float64 and affine.is_contiguous() and tuple(affine.shape) == (B, 12)
        d_aff = affine.data_ptr()
    try:
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        if topology is not None:
            ctx.voxelize_lattice_topo_dev(B, coords.data_ptr(), atom_offsets.data_ptr(), int(coords.shape[0]), topology, origins.data_ptr(), nv,
                                          float(voxelsize), d_box, int(max_images), out.data_ptr(), d_aff)
        else:
            ctx.voxelize_lattice_dev(B, coords.data_ptr(), atom_offsets.data_ptr(), int(coords.shape[0]),
                                     sigmas.data_ptr(), sigmas.dtype == torch.float64, C, origins.data_ptr(), nv,
                                     float(voxelsize), d_box, int(max_images), out.data_ptr(), d_aff)
    except BaseException:
        ctx.withdraw_promise()            # a promise made for THIS call (Context.promise_inputs) must not reach another one
        raise
    out = out.view(B, V, C)
    if channel_first:   # what the reference's tutorial builds by hand before nn.Conv3d
        return out.view(B, int(nv[0]), int(nv[1]), int(nv[2]), C).permute(0, 4, 1, 2, 3)
    return out


# ------------------------------------------------------------------------------------------------
# convenience wrappers in the reference's vocabulary
# ------------------------------------------------------------------------------------------------
def getVoxelDescriptorsBatch(coords_list, channels_list, centers, boxsize, voxelsize=1, boxes=None, ctx=None):
    """Batch analogue of ``getVoxelDescriptors(None, boxsize=..., center=..., usercoords=...,
    userchannels=...)``: item b is voxelized on a ``boxsize`` box centred on ``centers[b]``.

    Returns (features float32 [B,V,C], origins float64 [B,3], nvoxels int64 (3,)).
    """
    boxsize = np.array(boxsize, dtype=np.float64)
    centers = np.asarray(centers, dtype=np.float64).reshape(-1, 3)
    nvoxels = np.ceil(boxsize / voxelsize).astype(int)          # voxeldescriptors.py:242
    origins = centers - boxsize / 2                             # voxeldescriptors.py:243
    coords, sig, offs = pack_items(coords_list, channels_list)
    feats = voxelize_lattice(coords, offs, sig, origins, nvoxels, voxelsize, box=boxes, ctx=ctx)
    return feats, origins, nvoxels.astype(np.int64)


def voxelizeTrajectory(coords, channels, center, boxsize, voxelsize=1, box=None, frames=None, ctx=None):
    """All frames of a trajectory: coords float32 [N,3,F] (Molecule.coords), channels [N,C] sigma
    matrix shared by the frames, box float32 [3,F] (Molecule.box) or None.  Periodic frames use the
    orthorhombic minimum image of distance_utils.pyx:49-52 per frame.
    Returns (features float32 [F,V,C], origin float64 (3,), nvoxels).

    The frames go through ``iterVoxelizeTrajectory`` (slab copies into pinned staging, the transpose and the per-frame
    copies of the sigma matrix on the device) and every chunk's features are copied into the pre-touched result: the
    packed form below (``_voxelizeTrajectory_packed``: one host transpose of all coordinates and F host copies of the
    sigma matrix through the batched host call) took 244 ms for 256 frames of 30 000 atoms on a 48^3 grid, this 6 times
    less (tools/bench_voxelize_trajectory.py); bit-identical (tests/test_gpu_api.py)."""
    coords = np.asarray(coords)
    if coords.ndim != 3 or coords.shape[1] != 3:
        raise ValueError("coords must be (natoms, 3, nframes)")
    fr = np.arange(coords.shape[2]) if frames is None else np.asarray(frames)
    F = len(fr)
    sig = np.asarray(channels)
    boxsize_a = np.array(boxsize, dtype=np.float64)
    nvoxels = np.ceil(boxsize_a / voxelsize).astype(int)
    origin = np.asarray(center, dtype=np.float64) - boxsize_a / 2
    V, C = int(np.prod(nvoxels.astype(np.int64))), int(sig.shape[1])
    out = np.empty((F, V, C), dtype=np.float32)
    if F == 0 or V == 0:
        return out, origin, nvoxels.astype(np.int64)
    try:
        import torch
    except ImportError:                            # torch only owns device memory here: without it, the packed host call
        return _voxelizeTrajectory_packed(coords, channels, center, boxsize, voxelsize, box=box, frames=frames, ctx=ctx)
    # the GPU is the CONTEXT's (the caller's, or this thread's default: MKAMD_DEVICE / LOCAL_RANK), never torch's current one
    cctx = ctx or _lib.default_context()
    _lib._check(_lib.load().mkamd_prefault(_lib._ptr(out), out.nbytes))
    chunk = int(max(1, min(F, (256 << 20) // max(V * C * 4, 1))))      # ~256 MB of features per step
    pos = 0
    lib = _lib.load()
    for idx, feats in iterVoxelizeTrajectory(coords, sig, center, boxsize, voxelsize, box=box, frames=frames, chunk=chunk,
                                             device=f"cuda:{cctx.device}", ctx=cctx):
        n = len(idx)
        feats = feats.contiguous()
        torch.cuda.current_stream(feats.device).synchronize()
        # (the runtime's own device -> host copy into the pre-touched pages: 40 GB/s; a torch copy_ into a pageable
        #  tensor stages through bounce buffers at a third of that)
        _lib._check(lib.mkamd_copy_to_host(cctx._h, out[pos:pos + n].ctypes.data, feats.data_ptr(), n * V * C * 4))
        pos += n
    return out, origin, nvoxels.astype(np.int64)


def _voxelizeTrajectory_packed(coords, channels, center, boxsize, voxelsize=1, box=None, frames=None, ctx=None):
    """``voxelizeTrajectory`` through the batched HOST call: every frame a packed item (kept as the cross-check of the
    streamed form, and for callers without torch)."""
    coords = np.asarray(coords, dtype=np.float32)
    if coords.ndim != 3 or coords.shape[1] != 3:
        raise ValueError("coords must be (natoms, 3, nframes)")
    fr = np.arange(coords.shape[2]) if frames is None else np.asarray(frames)
    F, N = len(fr), coords.shape[0]
    packed = np.ascontiguousarray(np.transpose(coords[:, :, fr], (2, 0, 1))).reshape(F * N, 3)
    sig = np.asarray(channels)
    sig_all = np.ascontiguousarray(np.broadcast_to(sig[None], (F,) + sig.shape)).reshape(F * N, sig.shape[1])
    offs = np.arange(F + 1, dtype=np.int64) * N
    boxsize = np.array(boxsize, dtype=np.float64)
    nvoxels = np.ceil(boxsize / voxelsize).astype(int)
    origin = np.asarray(center, dtype=np.float64) - boxsize / 2
    bx = None
    if box is not None:
        bx = np.ascontiguousarray(np.asarray(box, dtype=np.float32)[:, fr].T)
    feats = voxelize_lattice(packed, offs, sig_all, np.broadcast_to(origin, (F, 3)), nvoxels, voxelsize,
                             box=bx, ctx=ctx)
    return feats, origin, nvoxels.astype(np.int64)


def _chunk_images(box3n, nvoxels, voxelsize) -> int:
    """Images per atom for one chunk's boxes ([3, n], Angstrom); raises for edges <= 10 A (zero boxes included)."""
    return max_images_per_atom(np.ascontiguousarray(np.asarray(box3n).T), nvoxels, voxelsize)


USE_TOPOLOGY = True        # the streamed voxelizers build a topology handle for their molecule (tests switch it off for the cross-check)
_PINNED = {}
_COPY_THREADS = 8
_POOL = None


def _copy_pool():
    """A few host threads for the strided trajectory-slab copy (numpy releases the GIL while copying)."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS)
    return _POOL


def _pinned_take(key, shape):
    """Pinned float32 staging tensor of `shape`: taken OUT of a small process-wide cache (pinning costs milliseconds)
    so that two streams running at once never share one; give it back with ``_pinned_give``."""
    import torch
    t = _PINNED.pop((key, tuple(shape)), None)
    return t if t is not None else torch.empty(tuple(shape), dtype=torch.float32).pin_memory()


def _pinned_give(key, t):
    for old in [q for q in _PINNED if q[0] == key]:          # one buffer per role is enough to keep
        del _PINNED[old]
    _PINNED[(key, tuple(t.shape))] = t


def chunk_plan(n_frames, chunk, ramp=0):
    """Boundaries of the chunks a streamed driver takes: equal chunks of ``chunk`` frames, or -- ``ramp`` > 0 -- a first chunk of
    ``ramp`` frames and every next one twice the size until ``chunk`` is reached.  A pipeline's first chunk is exposed (nothing
    runs beside its upload and decode) and its stages only overlap once several chunks are in flight, so a long trajectory is
    fed fastest by LARGE chunks that are approached through small ones (the device XTC walk takes ~10 ms beside the tile kernel
    whatever the chunk holds: below ~3 000 cfg4-sized frames per chunk it, not the voxelizer, sets the pace --
    docs/EXPERIMENTS_r6.md section 12)."""
    n_frames, chunk = int(n_frames), int(max(1, chunk))
    b, size = [0], int(ramp) if ramp and 0 < int(ramp) < chunk else chunk
    while b[-1] < n_frames:
        b.append(min(n_frames, b[-1] + size))
        size = min(chunk, 2 * size)
    return b


def _stream_voxelize(N, fr, fill, scale, has_box, channels, center, boxsize, voxelsize, chunk, device, channel_first, ctx,
                     max_images, fill_dev=None, pipelined=True, use_topology=None, ramp=0):
    """Core of the streamed voxelizers.  Host sources: ``fill(coords_np [N,3,n], box_np [3,n] | None, idx)`` produces chunk
    ``idx`` (frame indices) straight into pinned staging; device sources: ``fill_dev(xyz [n,N,3], box [n,3] | None, idx)``
    writes the chunk's frame-major device tensors itself.  Everything a chunk needs on the device -- the upload, the
    transpose to frame-major, the nm -> Angstrom ``scale`` (XTC stores nm), the box transpose -- is enqueued on a COPY
    stream into one of two slot-owned input buffers and ends in an event; the voxelize call of chunk k is handed that
    event as a promise (``Context.promise_inputs``, include/mkamd_voxel.h), so the library runs chunk k's binning pre-pass
    beside chunk k-1's tile kernel and nothing the consumer enqueues on its own stream between two chunks can sit in
    front of the inputs.  ``pipelined=False``: the same buffers, the compute stream waits for the event itself and the
    calls run in order (the cross-check of tests/test_gpu_api.py; bit-identical).

    Periodic boxes are checked per chunk on the host, where they are already at hand (``_chunk_images``: every edge
    > 10 A, images per atom recomputed from THIS chunk's boxes -- an NPT trajectory may shrink after its first frame);
    what only the device can see is polled without blocking after every chunk (``ctx.poll_errors``) and collected for
    good when the generator ends, so a bad frame raises instead of yielding silently incomplete features."""
    import torch

    # the device: the caller's, else the context's, else this thread's default context's (MKAMD_DEVICE / LOCAL_RANK) --
    # torch's current device only names the index of a bare "cuda"
    if device is None:
        dev = torch.device("cuda", (ctx or _lib.default_context()).device)
    else:
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
    if ctx is not None and ctx.device != dev.index:
        raise ValueError(f"ctx lives on GPU {ctx.device} but the tensors of this call would be allocated on {dev}: "
                         "pass a context of that device (kernels run on the context's device)")
    chunk = int(max(1, min(chunk, max(len(fr), 1))))
    sig = np.ascontiguousarray(channels)
    sig = sig.astype(np.float64 if sig.dtype == np.float64 else np.float32, copy=False)
    if sig.shape[0] != N:
        raise ValueError(f"channels has {sig.shape[0]} rows, the trajectory {N} atoms")
    boxsize = np.array(boxsize, dtype=np.float64)
    nvoxels = np.ceil(boxsize / voxelsize).astype(int)
    origin = np.asarray(center, dtype=np.float64) - boxsize / 2
    main = torch.cuda.current_stream(dev)
    # A HIGH-PRIORITY stream: what runs on it -- uploads, the transposing kernel, the device XTC decoder -- has to run BESIDE the
    # voxelizer.  Streams of one priority share a handful of hardware queues round-robin by creation order; when this stream landed on
    # the queue of the context's main or side stream, chunk k+1's decode and chunk k's tile kernel took turns instead of overlapping
    # (round 6: the XTC-fed leg flipped between 139 k and 193 k frames/s from one run to the next in ONE process).  High-priority
    # streams have queues of their own -- and a latency chain like the XTC walk wants to be dispatched first anyway.
    copy = torch.cuda.Stream(device=dev, priority=-1)
    host_source = fill_dev is None
    with torch.cuda.device(dev):
        # every frame has the molecule's sigmas: what the pre-pass derives from them is built ONCE (a topology handle,
        # include/mkamd_voxel.h (3c)) instead of per frame and call; molecules the handle does not take (more than 15 distinct
        # sigmas) -- and use_topology=False, the cross-check of the tests -- get the matrix repeated per frame as before
        topo = None
        run_ctx = ctx or _lib.default_context(dev.index)
        if use_topology is None:
            use_topology = USE_TOPOLOGY
        # (the A-B modes of a context -- the general path, a value tolerance -- are not served by topology calls)
        if use_topology and N > 0 and not getattr(run_ctx, "_force_general", False) and not getattr(run_ctx, "_value_tol", 0.0):
            try:
                topo = _lib.Topology(run_ctx, sig, float(voxelsize))
            except ValueError:
                topo = None
        d_sig = None if topo is not None else torch.as_tensor(sig, device=dev).repeat(chunk, 1).contiguous()      # [chunk*N, C]
        d_offs = torch.arange(chunk + 1, dtype=torch.int64, device=dev) * N
        d_org = torch.as_tensor(np.broadcast_to(origin, (chunk, 3)).copy(), device=dev)
        # two slots, each owning its pinned staging (host sources), its device slab and its frame-major inputs
        stage = [_pinned_take(("traj", i), (N * 3 * chunk,)) for i in range(2)] if host_source else None
        stage_box = [_pinned_take(("trajbox", i), (3 * chunk,)) for i in range(2)] if host_source else None
        d_slab = [torch.empty(N * 3 * chunk, dtype=torch.float32, device=dev) for _ in range(2)] if host_source else None
        d_slab_box = [torch.empty(3 * chunk, dtype=torch.float32, device=dev) for _ in range(2)] if host_source and has_box else None
        d_xyz = [torch.empty((chunk * N, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        d_bx = [torch.empty((chunk, 3), dtype=torch.float32, device=dev) for _ in range(2)] if has_box else [None, None]
        main.synchronize()                                        # the constants above (and a device-resident source, as far as
                                                                  # this stream produced it) are complete before any promise is made
        free = [torch.cuda.Event(), torch.cuda.Event()]          # staging buffer i may be overwritten (its H2D is done)
        ready = [torch.cuda.Event(), torch.cuda.Event()]         # the inputs of the chunk in slot i are complete
        consumed = [torch.cuda.Event(), torch.cuda.Event()]      # the call that read slot i's inputs is done (compute stream)
        images = [max_images, max_images]

        plan = chunk_plan(len(fr), chunk, ramp)

        def upload(k, slot):
            idx = fr[plan[k]:plan[k + 1]]
            n = len(idx)
            xyz = d_xyz[slot][:n * N].view(n, N, 3)
            if host_source:
                free[slot].synchronize()                          # the previous H2D out of this buffer is done
                hc = stage[slot][:N * 3 * n].view(N, 3, n)        # tight [N,3,n]: one contiguous H2D
                hb = stage_box[slot][:3 * n].view(3, n) if has_box else None
                fill(hc.numpy(), hb.numpy() if has_box else None, idx)
                images[slot] = max(max_images, _chunk_images(hb.numpy(), nvoxels, voxelsize)) if has_box else 1
            with torch.cuda.stream(copy):
                copy.wait_event(consumed[slot])                   # (never recorded yet for the first two chunks: no wait)
                if host_source:
                    slab = d_slab[slot][:N * 3 * n]
                    slab.copy_(hc.reshape(-1), non_blocking=True)
                    if has_box:
                        d_slab_box[slot][:3 * n].copy_(hb.reshape(-1), non_blocking=True)
                    free[slot].record(copy)
                    # frame-major and in Angstrom, by the library's transposing kernel on THIS stream (whole lines on both sides)
                    run_ctx.frames_to_items_dev(copy.cuda_stream, slab.data_ptr(), 3 * N, n, n, scale, xyz.data_ptr())
                    if has_box:
                        run_ctx.frames_to_items_dev(copy.cuda_stream, d_slab_box[slot].data_ptr(), 3, n, n, 1.0, d_bx[slot].data_ptr())
                else:
                    images[slot] = fill_dev(copy, xyz, d_bx[slot][:n] if has_box else None, idx) or max_images
                ready[slot].record(copy)
            return idx

        try:
            nchunks = len(plan) - 1
            pending = upload(0, 0) if nchunks else None
            for k in range(nchunks):
                slot = k & 1
                idx = pending
                if k + 1 < nchunks:
                    pending = upload(k + 1, slot ^ 1)
                n = len(idx)
                if pipelined:
                    run_ctx.promise_inputs(ready[slot])           # the library waits for it where the pre-pass runs
                else:
                    main.wait_event(ready[slot])
                feats = voxelize_lattice_torch(d_xyz[slot][:n * N], d_offs[:n + 1], None if topo is not None else d_sig[:n * N], d_org[:n], nvoxels,
                                               voxelsize, box=d_bx[slot][:n] if has_box else None, max_images=images[slot], ctx=run_ctx,
                                               channel_first=channel_first, topology=topo)
                consumed[slot].record(torch.cuda.current_stream(dev))
                run_ctx.poll_errors()                             # non-blocking: errors of the chunks already finished
                yield idx, feats
            run_ctx.synchronize()                                 # the last chunks' asynchronous errors, if any
        finally:                                                  # also when the consumer stops early
            copy.synchronize()                                    # no H2D still reading the staging buffers
            run_ctx.withdraw_promise()                            # (a consumer that stopped between a promise and its call)
            if host_source:
                for i in range(2):
                    _pinned_give(("traj", i), stage[i])
                    _pinned_give(("trajbox", i), stage_box[i])


def iterVoxelizeTrajectory(coords, channels, center, boxsize, voxelsize=1, box=None, frames=None, chunk=512,
                           device=None, channel_first=False, ctx=None, pipelined=True):
    """Stream a trajectory through the GPU chunk by chunk ("trajectory feeding"): yields
    ``(frame_indices, features)`` with ``features`` a float32 CUDA tensor ``[n, V, C]`` (or ``[n, C, nx, ny, nz]`` with
    ``channel_first``) for ``n <= chunk`` frames at a time.

    ``coords`` is ``Molecule.coords`` (float32 ``[N, 3, F]``, frame fastest) on the host -- or the same array as a CUDA
    tensor when the trajectory already lives in HBM (a simulation engine's output, a previous stage) --, ``box``
    ``Molecule.box`` (``[3, F]``) or None.  Host source: per chunk the host only copies the ``[N, 3, n]`` slab into one of
    two pinned staging buffers (a few host threads; runs of ``n`` contiguous floats); the transpose to frame-major happens
    on the device.  A copy stream prepares chunk k+1 while chunk k is voxelized, and the library overlaps chunk k+1's
    binning pre-pass with chunk k's tile kernel (``_stream_voxelize``), so the consumer (a model, a reduction) sees a
    steady feed whose rate is the slower of PCIe and the voxelizer.  The tensors are yours to keep: each chunk gets
    fresh memory.
    """
    on_device = hasattr(coords, "is_cuda") and bool(coords.is_cuda)
    if not on_device:
        coords = np.asarray(coords)
        if coords.dtype != np.float32:
            coords = coords.astype(np.float32)
    if coords.ndim != 3 or coords.shape[1] != 3:
        raise ValueError("coords must be (natoms, 3, nframes)")
    N = int(coords.shape[0])
    fr = np.arange(coords.shape[2]) if frames is None else np.asarray(frames, dtype=np.int64)
    contiguous = frames is None or (len(fr) > 0 and np.array_equal(fr, np.arange(fr[0], fr[0] + len(fr))))
    max_images = 1
    box_t = None
    if box is not None:
        if hasattr(box, "is_cuda"):
            box_t = box if box.is_cuda else None
            box = box.detach().cpu().numpy()
        box = np.asarray(box, dtype=np.float32)
        nvoxels = np.ceil(np.array(boxsize, dtype=np.float64) / voxelsize).astype(int)
        max_images = max_images_per_atom(np.ascontiguousarray(box[:, fr].T), nvoxels, voxelsize)

    def fill(dst, dst_box, idx):
        n = len(idx)
        if contiguous:                                            # strided slab -> pinned, split over a few host threads
            f0 = int(idx[0])
            parts = np.linspace(0, N, _COPY_THREADS + 1).astype(int)
            list(_copy_pool().map(lambda ab: np.copyto(dst[ab[0]:ab[1]], coords[ab[0]:ab[1], :, f0:f0 + n]),
                                  zip(parts[:-1], parts[1:])))
            if dst_box is not None:
                np.copyto(dst_box, box[:, f0:f0 + n])
        else:
            np.copyto(dst, coords[:, :, idx])
            if dst_box is not None:
                np.copyto(dst_box, box[:, idx])

    fill_dev = None
    if on_device:
        import torch
        if coords.dtype != torch.float32:
            raise ValueError("a device-resident trajectory must be float32")
        if device is None:
            device = coords.device
        if box is not None and box_t is None:
            box_t = torch.as_tensor(box, device=coords.device)

        cctx = ctx or _lib.default_context(coords.device.index if coords.device.index is not None else 0)
        F_all = int(coords.shape[2])
        if not coords.is_contiguous():
            coords = coords.contiguous()
        if box_t is not None:
            box_t = box_t.to(torch.float32).contiguous()

        def fill_dev(copy, xyz, bx, idx):                         # inside the copy-stream context of _stream_voxelize
            n = len(idx)
            if contiguous:                                        # a window of the resident array: the library's transposing kernel
                f0 = int(idx[0])
                cctx.frames_to_items_dev(copy.cuda_stream, coords.data_ptr() + 4 * f0, 3 * N, F_all, n, 1.0, xyz.data_ptr())
                if bx is not None:
                    cctx.frames_to_items_dev(copy.cuda_stream, box_t.data_ptr() + 4 * f0, 3, F_all, n, 1.0, bx.data_ptr())
            else:
                sel = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=coords.device)
                xyz.copy_(coords.index_select(2, sel).permute(2, 0, 1))
                if bx is not None:
                    bx.copy_(box_t.index_select(1, sel).t())
            return _chunk_images(box[:, idx], nvoxels, voxelsize) if box is not None else 1

    yield from _stream_voxelize(N, fr, fill, 1.0, box is not None, channels, center, boxsize, voxelsize, chunk, device,
                                channel_first, ctx, max_images, fill_dev=fill_dev, pipelined=pipelined)


def iterVoxelizeXTC(filename, channels, center, boxsize, voxelsize=1, pbc=True, frames=None, chunk=1024, device=None,
                    channel_first=False, ctx=None, nthreads=0, pipelined=True, decode="auto", ramp=0):
    """``iterVoxelizeTrajectory`` fed straight from an XTC file.  ``pbc``: use the frames' box (orthorhombic lengths of the
    box vectors) for the minimum image.  Coordinates are converted from the file's nm to Angstrom on the device, like
    ``readers.XTCread`` does on the host.  ``chunk``: frames per step of the pipeline; ``ramp`` > 0: the first step takes
    ``ramp`` frames and every next one twice as many until ``chunk`` is reached (``chunk_plan``: for long trajectories
    ``chunk=4096, ramp=512`` feeds fastest -- the yielded batches then differ in size).

    ``decode``: where the coordinates are decompressed --
      ``"gpu"``   on the device (csrc/xtc_gpu.h): per chunk the host parses the record headers and copies the records' BYTES
                  (~5 per atom) into pinned staging; a lane walks each frame's bit stream, a thread decodes each group of
                  atoms, straight into the voxelizer's frame-major items -- the same bits as the host decoder.  The walk
                  takes about the same time for 256 frames as for 4 096 (a wave per 64 frames, most of the chip idle), so
                  LARGE chunks feed fastest: 1 024 frames of 30 000 atoms decode in 3.1 ms (their voxelization takes 4.3).
      ``"host"``  by host threads (libmkamd.so's decoder, ``moleculekit_amd.xtc``) into pinned staging, in blocks of 16
                  frames per thread: bound by the host's cores (28 k frames/s of 30 000 atoms on the 16 an MI355X box grants).
      ``"auto"``  the device for a contiguous ascending range of frames whose headers it can take (``xtc.device_decodable``:
                  no coordinate packed into more than 64 bits, < 2^21 atoms), the host otherwise (a sparse selection would
                  copy the whole span of the file between its first and last frame).
    A frame the device decoder refuses after all, or a corrupt one, raises -- at the latest when the generator ends (like the
    voxelizer's own asynchronous errors)."""
    import ctypes

    from . import xtc as _xtc
    if decode not in ("auto", "gpu", "host"):
        raise ValueError('decode must be "auto", "gpu" or "host"')
    natoms, nframes = _xtc.get_xtc_natoms(filename), _xtc.get_xtc_nframes(filename)
    fr = np.arange(nframes, dtype=np.int64) if frames is None else np.asarray(frames, dtype=np.int64)
    lib, path = _lib.load(), _xtc._path(filename)
    max_images = 1
    nvoxels = np.ceil(np.array(boxsize, dtype=np.float64) / voxelsize).astype(int)
    if pbc and len(fr):
        _, bv, _, _ = _xtc.read_xtc_frames(filename, fr[:1])     # first guess from the first frame's box; every chunk
                                                                 # recomputes it from its own boxes (_stream_voxelize)
        lengths = np.sqrt((bv[:, :, 0].astype(np.float64) ** 2).sum(axis=1)) * 10.0
        if not np.all(lengths > 0):
            raise ValueError("pbc=True but the XTC frames carry no box")
        max_images = max_images_per_atom(lengths[None, :] * 0.98, nvoxels, voxelsize)    # 2 % slack for box fluctuations

    def box_lengths(bv):                                          # Angstrom, float32 like the reference's conversion
        bv = bv * np.float32(10.0)                                # (readers.py:1848-1859)
        return np.sqrt(np.sum(bv * bv, axis=1))                   # [3, n]

    if decode == "auto":
        contiguous = len(fr) > 0 and np.array_equal(fr, np.arange(fr[0], fr[0] + len(fr)))
        decode = "gpu" if contiguous and _xtc.device_decodable(_xtc.chunk_desc(filename, fr[:1], natoms)[0], natoms) else "host"
    if decode == "gpu" and len(fr):
        yield from _iter_xtc_gpu(filename, path, natoms, fr, box_lengths, nvoxels, bool(pbc), channels, center, boxsize, voxelsize,
                                 chunk, device, channel_first, ctx, max_images, int(nthreads), pipelined, ramp)
        return

    def fill(dst, dst_box, idx):
        n = len(idx)
        sel = np.ascontiguousarray(idx, dtype=np.int64)
        bv = np.empty((3, 3, n), dtype=np.float32)
        t = np.empty(n, dtype=np.float32)
        st = np.empty(n, dtype=np.int32)
        _lib._check(lib.mkamd_xtc_read(path, _lib._ptr(sel), n, natoms, _lib._ptr(dst), _lib._ptr(bv), _lib._ptr(t),
                                       _lib._ptr(st), int(nthreads)))
        if dst_box is not None:
            np.copyto(dst_box, box_lengths(bv))

    yield from _stream_voxelize(natoms, fr, fill, 10.0, bool(pbc), channels, center, boxsize, voxelsize, chunk, device,
                                channel_first, ctx, max_images, pipelined=pipelined, ramp=ramp)


def _iter_xtc_gpu(filename, path, natoms, fr, box_lengths, nvoxels, has_box, channels, center, boxsize, voxelsize, chunk, device,
                  channel_first, ctx, max_images, nthreads, pipelined, ramp=0):
    """``iterVoxelizeXTC(decode="gpu")``: the chunk source of ``_stream_voxelize`` that decodes on the device.  Per chunk, on
    the host: ``mkamd_xtc_chunk_desc`` (headers -> descriptors, box vectors) and ``mkamd_xtc_copy_bytes`` (the records into
    one of two pinned byte buffers); on an UPLOAD stream the records' H2D (beside the previous chunk's decode); on the copy
    stream the descriptors, ``mkamd_xtc_decode_dev`` into the slot's frame-major items, the statuses back into pinned memory.

    The host never waits for a decode it needs soon: a byte buffer is reused when ITS upload is done (the device side orders
    the upload behind the decode that still reads the device copy), and the small pinned buffers (descriptors, boxes,
    statuses) rotate over four chunks, so the host blocks on the decode of chunk k-4 at chunk k -- it prepares chunk k while
    the device still decodes k-1 and k-2 (with a wait on chunk k-2 the feed ran at decode + host + upload per two chunks:
    10.6 ms per 2 048 frames instead of 9.5).  Statuses are looked at without blocking as chunks complete, and for good at the
    end."""
    import torch

    from . import xtc as _xtc
    lib = _lib.load()
    run_ctx = ctx or _lib.default_context(None if device is None else torch.device(device).index)
    dev = torch.device("cuda", run_ctx.device)
    chunk = int(max(1, min(chunk, len(fr))))
    NS = 4
    state = {"k": 0}
    with torch.cuda.device(dev):
        h_raw, d_raw = [None, None], [None, None]                # byte buffers, grown as chunks need them (pinning hundreds of
                                                                 # MB costs 40-80 ms: the pinned ones are kept between calls)
        for i in range(2):
            for key in [q for q in _PINNED if q[0] == ("xtcraw", i)]:
                h_raw[i] = _PINNED.pop(key)
        h_desc = [torch.empty((chunk, _xtc.DESC_BYTES), dtype=torch.uint8, pin_memory=True) for _ in range(NS)]
        h_box = [torch.empty((chunk, 3), dtype=torch.float32, pin_memory=True) for _ in range(NS)]
        h_st = [torch.zeros(chunk, dtype=torch.int32).pin_memory() for _ in range(NS)]
        d_desc = [torch.empty((chunk, _xtc.DESC_BYTES), dtype=torch.uint8, device=dev) for _ in range(2)]
        d_st = [torch.empty(chunk, dtype=torch.int32, device=dev) for _ in range(2)]
        work = torch.empty(int(lib.mkamd_xtc_decode_work_bytes(chunk, natoms)), dtype=torch.uint8, device=dev)  # (decodes are
        h2d = torch.cuda.Stream(device=dev)                      #  in order on one stream: one work buffer)
        up = [torch.cuda.Event(), torch.cuda.Event()]            # the upload out of h_raw[slot] is done
        done = [torch.cuda.Event() for _ in range(NS)]           # chunk k's decode and status copy are done (k % NS)
    in_flight = [None] * NS                                       # frame indices whose statuses h_st[k % NS] will hold

    def check(ss, block):
        idx = in_flight[ss]
        if idx is None or not (block or done[ss].query()):
            return
        if block:
            done[ss].synchronize()
        in_flight[ss] = None
        st = h_st[ss][:len(idx)].numpy()
        if st.any():
            bad = int(np.flatnonzero(st)[0])
            raise RuntimeError(f"{filename}: frame {int(idx[bad])} " + ("is corrupt" if st[bad] == 1 else "is outside what the device "
                               'decoder takes (a number of more than 64 bits): read this file with decode="host"'))

    def fill_dev(copy, xyz, bx, idx):
        k = state["k"]
        state["k"] += 1
        slot, ss = k & 1, k % NS
        check(ss, True)                                           # chunk k-4: long done; frees the small pinned buffers
        up[slot].synchronize()                                    # chunk k-2's upload: h_raw[slot] may be overwritten
        n = len(idx)
        # the records' bytes first (their range comes from the frame index alone), then the headers out of the COPY: read through a fresh
        # mapping of the file every header costs a page fault -- 2 ms per 2 048 frames of a host side that is the feed's pace (round 6)
        lo, hi = _xtc.byte_range(filename, idx, natoms)
        need = hi - lo + _xtc.XTC_PAD
        if h_raw[slot] is None or h_raw[slot].numel() < need:
            h_raw[slot] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        if d_raw[slot] is None or d_raw[slot].numel() < need:
            done[(k - 2) % NS].synchronize()                      # (nobody reads the device copy that is let go)
            with torch.cuda.stream(h2d):
                d_raw[slot] = torch.empty(h_raw[slot].numel(), dtype=torch.uint8, device=dev)
        _lib._check(lib.mkamd_xtc_copy_bytes(path, lo, hi, h_raw[slot].data_ptr(), nthreads))
        h_raw[slot][hi - lo:need].zero_()                         # (the read-ahead pad: never used, but not left to chance)
        desc, lo_d, hi_d, bv, _, _ = _xtc.chunk_desc_mem(filename, idx, natoms, h_raw[slot].data_ptr(), lo, hi)
        if lo_d != lo or hi_d > hi:                               # (the descriptors' offsets count from lo_d: the copy must start there)
            raise RuntimeError(f"{filename}: the frame index and the record headers disagree about the byte range of frames {int(idx[0])}..{int(idx[-1])}")
        h_desc[ss][:n].numpy()[...] = desc
        images = max_images
        if has_box:
            hb = box_lengths(bv)                                  # [3, n]
            h_box[ss][:n].numpy()[...] = hb.T
            images = max(max_images, _chunk_images(hb, nvoxels, voxelsize))
        with torch.cuda.stream(h2d):
            if k >= 2:
                h2d.wait_event(done[(k - 2) % NS])                # the decode that still reads d_raw[slot]
            d_raw[slot][:need].copy_(h_raw[slot][:need], non_blocking=True)
            up[slot].record(h2d)
        copy.wait_event(up[slot])
        d_desc[slot][:n].copy_(h_desc[ss][:n], non_blocking=True)
        if has_box:
            bx.copy_(h_box[ss][:n], non_blocking=True)
        _lib._check(lib.mkamd_xtc_decode_dev(run_ctx._h, copy.cuda_stream or None, d_raw[slot].data_ptr(), d_desc[slot].data_ptr(), n,
                                             natoms, 10.0, xyz.data_ptr(), d_st[slot].data_ptr(), work.data_ptr(), work.numel()))
        h_st[ss][:n].copy_(d_st[slot][:n], non_blocking=True)
        done[ss].record(copy)
        in_flight[ss] = np.array(idx, copy=True)
        return images

    gen = _stream_voxelize(natoms, fr, None, 10.0, has_box, channels, center, boxsize, voxelsize, chunk, device, channel_first,
                           run_ctx, max_images, fill_dev=fill_dev, pipelined=pipelined, ramp=ramp)
    try:
        for item in gen:
            for ss in range(NS):
                check(ss, False)
            yield item
        for ss in range(NS):
            check(ss, True)
    finally:
        gen.close()                                               # (synchronizes the copy stream, which waited for the uploads)
        h2d.synchronize()
        for i in range(2):
            if h_raw[i] is not None:
                _pinned_give(("xtcraw", i), h_raw[i])
