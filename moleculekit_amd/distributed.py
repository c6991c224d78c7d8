"""Multi-GPU sharding of a batch of independent voxelization items.

Every molecule / pose / frame is an independent unit, so the batch is partitioned contiguously
across ranks (one process per GPU, ``torch.distributed``; backend ``nccl`` is RCCL over xGMI on
ROCm) with NO collective on the compute path.  The only exchange is the optional, trivial gather
of the per-rank feature tensors at the end (``gather_features``): an RCCL all-gather of equally
sized (padded) shards, or gather-to-root.

The reference has no distributed code at all (no NCCL/MPI call sites); this
module is new, not a translation.
"""
from __future__ import annotations

import os

import numpy as np


def world():
    """(rank, world_size) from torch.distributed if initialised, else from the torchrun env."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n_items: int, world_size: int, weights=None):
    """Contiguous partition of ``range(n_items)`` into ``world_size`` shards.

    Returns int64 [world_size+1] boundaries.  Without ``weights`` shard sizes differ by at most one;
    with ``weights`` (e.g. atoms per item -- cfg5's molecules vary in size) the boundaries balance the
    cumulative weight instead (each item still goes to exactly one rank, order preserved).
    """
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    if weights is None:
        base, rem = divmod(int(n_items), world_size)
        sizes = np.full(world_size, base, dtype=np.int64)
        sizes[:rem] += 1
        return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    w = np.asarray(weights, dtype=np.float64)
    if w.shape != (n_items,):
        raise ValueError("weights must have one entry per item")
    cum = np.concatenate([[0.0], np.cumsum(w)])
    targets = cum[-1] * np.arange(1, world_size) / world_size
    inner = np.searchsorted(cum, targets, side="left")
    b = np.concatenate([[0], inner, [n_items]]).astype(np.int64)
    return np.maximum.accumulate(b)


def shard_packed(coords, atom_offsets, sigmas, origins, box, lo, hi):
    """Slice packed batch arrays down to items [lo, hi)."""
    a0, a1 = int(atom_offsets[lo]), int(atom_offsets[hi])
    offs = np.asarray(atom_offsets[lo:hi + 1], dtype=np.int64) - a0
    return (coords[a0:a1], offs, sigmas[a0:a1], origins[lo:hi], None if box is None else box[lo:hi])


def gather_features(local, bounds, group=None, dst=None):
    """Collect the per-rank feature shards ``local`` [B_local, V, C] (torch tensor on this rank's
    device) into the full [B, V, C] tensor.

    Shards may differ in size by the partition; they are padded to the largest shard so that one
    ``all_gather_into_tensor`` (a single RCCL ring all-gather over xGMI) moves everything, then the
    padding is dropped.  With ``dst`` set, only that rank receives (``gather``); others get None.
    """
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = np.diff(np.asarray(bounds))
    assert len(sizes) == ws and local.shape[0] == sizes[rank]
    bmax = int(sizes.max()) if ws else 0
    tail = tuple(local.shape[1:])
    padded = local
    if local.shape[0] != bmax:
        padded = torch.zeros((bmax,) + tail, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    if dst is None:
        full = torch.empty((ws * bmax,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, padded, group=group)
        parts = [full[r * bmax: r * bmax + int(sizes[r])] for r in range(ws)]
        return torch.cat(parts, dim=0) if any(int(s) != bmax for s in sizes) else full
    recv = [torch.empty_like(padded) for _ in range(ws)] if rank == dst else None
    dist.gather(padded, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][: int(sizes[r])] for r in range(ws)], dim=0)


def voxelize_sharded(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None, gather=True,
                     compute=None, device=None, balance_by_atoms=True):
    """Voxelize a (host-resident, packed) batch across all ranks of the default process group.

    Every rank passes the SAME full batch description; rank r computes only its contiguous shard on
    its own GPU and -- when ``gather`` -- every rank returns the full float32 [B, V, C] tensor,
    otherwise its local shard (what a data-parallel trainer wants).

    ``compute(coords, offs, sigmas, origins, nvoxels, voxelsize, box) -> torch.Tensor`` defaults to
    the HIP path (``batch.voxelize_lattice_torch`` on this rank's device); the CPU test-suite injects
    the oracle here to exercise the sharding / gather logic under ``gloo``.
    Returns (features, bounds).
    """
    import torch

    rank, ws = world()
    B = len(atom_offsets) - 1
    weights = np.diff(np.asarray(atom_offsets)) + 1.0 if balance_by_atoms else None
    bounds = shard_bounds(B, ws, weights)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    c, offs, s, o, bx = shard_packed(np.asarray(coords), np.asarray(atom_offsets), np.asarray(sigmas),
                                     np.asarray(origins, dtype=np.float64).reshape(-1, 3),
                                     None if box is None else np.asarray(box), lo, hi)
    if compute is None:
        from . import batch

        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) if device is None else device)

        def compute(c, offs, s, o, nv, vs, bx):
            mi = 1 if bx is None else batch.max_images_per_atom(bx, nv, vs)
            t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
            return batch.voxelize_lattice_torch(
                t(c, np.float32).reshape(-1, 3), t(offs, np.int64), t(s, np.float32), t(o, np.float64), nv, vs,
                box=None if bx is None else t(bx, np.float32), max_images=mi)

    local = compute(c, offs, s, o, nvoxels, voxelsize, bx)
    if gather and ws > 1:
        return gather_features(local, bounds), bounds
    return local, bounds
