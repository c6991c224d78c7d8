"""Multi-GPU sharding of a batch of independent voxelization items.

Every molecule / pose / frame is an independent unit, so the batch is partitioned contiguously
across ranks (one process per GPU, ``torch.distributed``; backend ``nccl`` is RCCL over xGMI on
ROCm) with NO collective on the compute path.  The only exchange is the optional, trivial gather
of the per-rank feature tensors: gathering *all* features is xGMI-link
bound, so it is offered
  * not at all          -- ``ShardedVoxelizer.voxelize``: features stay sharded (what a data-parallel
                           consumer wants; this is what ``bench.py --gpus N`` times),
  * after the compute   -- ``gather_features``: one padded RCCL all-gather (or gather-to-root),
  * overlapped          -- ``ShardedVoxelizer.voxelize_gather``: the shard is voxelized in chunks and
                           chunk k travels on a communication stream while chunk k+1 is computed; equal shards are
                           computed straight into the result and exchanged point to point (all seven xGMI links of a
                           GPU busy at once, no staging copy).

Data path of one rank: it holds ONLY its own shard -- host arrays are sliced (or produced by a
``loader(lo, hi)`` callback, so no rank ever materialises the whole batch), staged through pinned
memory, uploaded once on a copy stream and kept resident in HBM; every later call works on the
resident shard.

The reference has no distributed code at all (no NCCL/MPI call sites); this
module is new, not a translation.
"""
from __future__ import annotations

import os

import numpy as np


P2P_MAX_BYTES = 1 << 30        # largest single point-to-point message voxelize_gather posts (see there)


def world(group=None):
    """(rank, world_size) from torch.distributed if initialised, else from the torchrun env."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
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
    """Slice packed batch arrays down to items [lo, hi) (views; the offsets are rebased)."""
    a0, a1 = int(atom_offsets[lo]), int(atom_offsets[hi])
    offs = np.asarray(atom_offsets[lo:hi + 1], dtype=np.int64) - a0
    return (coords[a0:a1], offs, sigmas[a0:a1], origins[lo:hi], None if box is None else box[lo:hi])


def chunk_bounds(n_local: int, nchunks: int):
    """Split a shard of ``n_local`` items into ``nchunks`` contiguous chunks (sizes differ by at most one;
    trailing chunks may be empty).  Every rank uses the same ``nchunks`` so that the per-chunk collectives match."""
    return shard_bounds(n_local, max(int(nchunks), 1))


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
        # equal shards (the weak-scaling case): the collective writes every shard at its final rows, nothing is copied after
        full = torch.empty((ws * bmax,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, padded, group=group)
        if all(int(s) == bmax for s in sizes):
            return full
        parts = [full[r * bmax: r * bmax + int(sizes[r])] for r in range(ws)]
        return torch.cat(parts, dim=0)
    recv = [torch.empty_like(padded) for _ in range(ws)] if rank == dst else None
    dist.gather(padded, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][: int(sizes[r])] for r in range(ws)], dim=0)


class ShardedVoxelizer:
    """This rank's device-resident shard of a batch of lattice-grid items, and the ways to get features out of it.

    Build with ``from_host`` (every rank passes the same packed host arrays; only views of this rank's shard are
    touched) or ``from_loader`` (``loader(lo, hi)`` produces items [lo, hi) -- no rank holds the whole batch).
    ``compute(coords, offs, sigmas, origins, nvoxels, voxelsize, box) -> torch.Tensor`` defaults to the HIP path
    (``batch.voxelize_lattice_torch`` on this rank's GPU); the CPU test-suite injects a stand-in to exercise the
    sharding / staging / gather logic under ``gloo`` -- the product never does.
    """

    def __init__(self, n_items, bounds, shard, nvoxels, voxelsize, device=None, compute=None, group=None, ctx=None,
                 pipelined=True, shared_sigmas=False):
        import torch

        # shared_sigmas: the items are sets of coordinates of ONE molecule -- the frames of a trajectory (cfg4) --, so everything the
        # pre-pass derives from the sigmas is built once (a topology handle, include/mkamd_voxel.h (3c)) and only the molecule's own
        # sigma matrix lives on the device.  Checked here, on the host: every item the same length, every item's rows identical.

        # The shard's buffers belong to this object and never change after the upload below has completed: every voxelize
        # call over them is made with the library's per-call promise (Context.promise_inputs, include/mkamd_voxel.h), so
        # the binning pre-pass of one call runs beside the tile kernel of the previous one -- back-to-back `voxelize()`
        # calls and the chunks of `voxelize_gather` alike.  `pipelined = False` runs the calls in order (bit-identical).
        self.pipelined = bool(pipelined)
        self.group = group
        self.rank, self.world = world(group)
        self.n_items = int(n_items)
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.lo, self.hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.nvoxels = np.ascontiguousarray(nvoxels, dtype=np.int32).reshape(3)
        self.V = int(np.prod(self.nvoxels.astype(np.int64)))
        self.voxelsize = float(voxelsize)
        self._compute = compute
        self._ctx = ctx
        coords, offs, sigmas, origins, box = shard
        self.n_local = int(len(offs) - 1)
        assert self.n_local == self.hi - self.lo, "the shard does not match this rank's bounds"
        self.C = int(np.asarray(sigmas).shape[1]) if np.asarray(sigmas).ndim == 2 else 0
        self._offs_host = np.ascontiguousarray(offs, dtype=np.int64)
        self.max_images = 1
        if compute is None:
            if device is None:
                device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", self.rank)) % max(torch.cuda.device_count(), 1))
            self.device = torch.device(device)
            if self.device.type != "cuda":
                raise RuntimeError("ShardedVoxelizer needs a HIP device (there is no CPU path); tests inject `compute`")
            if box is not None and len(box):
                from . import batch
                self.max_images = batch.max_images_per_atom(box, self.nvoxels, self.voxelsize)
        else:
            self.device = torch.device("cpu" if device is None else device)
        sig = np.asarray(sigmas)
        sig_dt = np.float64 if sig.dtype == np.float64 else np.float32
        self._topo = None
        self._shared = False
        if shared_sigmas and self.n_local > 0:
            # (checked whoever computes: an injected `compute` -- the CPU test path -- gets the matrix repeated per item, as it
            #  slices rows by atom; round 5 skipped the check there and would have sliced a molecule's [n, C] matrix wrongly)
            sizes = np.diff(self._offs_host)
            n0 = int(sizes[0])
            if n0 > 0 and np.all(sizes == n0):
                if sig.shape[0] == n0 * self.n_local:                  # the matrix repeated per item: it has to BE a repeat
                    rows = sig.reshape(self.n_local, n0, -1)
                    if not all(np.array_equal(rows[0], r) for r in rows[1:]):
                        raise ValueError("shared_sigmas: the items' sigma rows differ (not frames of one molecule)")
                    sig = np.ascontiguousarray(rows[0])
                elif sig.shape[0] != n0:
                    raise ValueError("shared_sigmas: pass the molecule's [n, C] matrix or its repeat per item")
                if compute is None:
                    self._shared = True
                else:
                    sig = np.tile(sig, (self.n_local, 1))
            else:
                raise ValueError("shared_sigmas: every item must have the molecule's atom count")
        self._d = self._upload(dict(coords=(coords, np.float32), offs=(self._offs_host, np.int64), sigmas=(sig, sig_dt),
                                    origins=(np.asarray(origins, dtype=np.float64).reshape(-1, 3), np.float64),
                                    box=(box, np.float32)))
        self._chunk_offs = {}
        if self._shared:
            from . import _lib
            tctx = self._ctx or _lib.default_context(self.device.index if self.device.index is not None else 0)
            if not getattr(tctx, "_force_general", False) and not getattr(tctx, "_value_tol", 0.0):
                torch.cuda.current_stream(self.device).synchronize()
                try:
                    self._topo = _lib.Topology(tctx, self._d["sigmas"], self.voxelsize)
                except ValueError:                                     # more than 15 distinct sigmas: the plain call on the repeated matrix
                    self._topo = None
            if self._topo is None:
                self._d["sigmas"] = self._d["sigmas"].repeat(self.n_local, 1).contiguous()
                self._shared = False

    # ---- construction -------------------------------------------------------------------------------------
    @classmethod
    def from_host(cls, coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None, balance_by_atoms=True, **kw):
        rank, ws = world(kw.get("group"))
        atom_offsets = np.asarray(atom_offsets)
        B = len(atom_offsets) - 1
        weights = np.diff(atom_offsets) + 1.0 if balance_by_atoms else None
        bounds = shard_bounds(B, ws, weights)
        shard = shard_packed(np.asarray(coords), atom_offsets, np.asarray(sigmas),
                             np.asarray(origins, dtype=np.float64).reshape(-1, 3),
                             None if box is None else np.asarray(box), int(bounds[rank]), int(bounds[rank + 1]))
        return cls(B, bounds, shard, nvoxels, voxelsize, **kw)

    @classmethod
    def from_loader(cls, n_items, loader, nvoxels, voxelsize, weights=None, **kw):
        """``loader(lo, hi) -> (coords [n,3], atom_offsets [hi-lo+1] starting at 0, sigmas [n,C], origins [hi-lo,3],
        box [hi-lo,3] | None)`` for items [lo, hi); ``weights`` (e.g. atoms per item) balance the partition."""
        rank, ws = world(kw.get("group"))
        bounds = shard_bounds(int(n_items), ws, weights)
        return cls(n_items, bounds, tuple(loader(int(bounds[rank]), int(bounds[rank + 1]))), nvoxels, voxelsize, **kw)

    def _upload(self, arrays):
        """Host arrays -> device tensors through pinned staging on a copy stream (one H2D per array, asynchronous;
        the compute stream waits for the copies' event).  On the CPU test path: plain tensors."""
        import torch

        out = {}
        if self.device.type != "cuda":
            for k, (a, dt) in arrays.items():
                out[k] = None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dt))
            return out
        copy = torch.cuda.Stream(device=self.device)
        keep = []
        with torch.cuda.stream(copy):
            for k, (a, dt) in arrays.items():
                if a is None:
                    out[k] = None
                    continue
                a = np.asarray(a)
                pinned = torch.empty(a.shape, dtype=getattr(torch, np.dtype(dt).name), pin_memory=True)
                np.copyto(pinned.numpy(), a, casting="same_kind")        # the only host-side pass over the shard
                out[k] = pinned.to(self.device, non_blocking=True)
                keep.append(pinned)
        ev = torch.cuda.Event()
        ev.record(copy)
        torch.cuda.current_stream(self.device).wait_event(ev)
        for t in out.values():
            if t is not None:
                t.record_stream(torch.cuda.current_stream(self.device))
        copy.synchronize()              # the pinned staging buffers are released here (set-up, not the hot path)
        return out

    # ---- compute ------------------------------------------------------------------------------------------
    def _run(self, coords, offs, sigmas, origins, box, out=None):
        if self._compute is not None:
            t = lambda x: None if x is None else x.numpy()
            res = self._compute(t(coords), t(offs), t(sigmas), t(origins), self.nvoxels, self.voxelsize, t(box))
            if out is not None:
                out.copy_(res)
                return out
            return res
        from . import _lib, batch

        ctx = self._ctx or _lib.default_context(self.device.index if self.device.index is not None else 0)
        if self.pipelined:
            ctx.promise_inputs(None)          # resident since _upload's host-side wait: complete, and nobody writes them
        return batch.voxelize_lattice_torch(coords, offs, None if self._topo is not None else sigmas, origins, self.nvoxels, self.voxelsize,
                                            box=box, max_images=self.max_images, out=out, ctx=ctx, topology=self._topo)

    def _items(self, lo, hi):
        """Views of the resident shard for local items [lo, hi) (the rebased offsets are built once per range)."""
        import torch

        d = self._d
        a0, a1 = int(self._offs_host[lo]), int(self._offs_host[hi])
        if (lo, hi) == (0, self.n_local):
            offs = d["offs"]
        else:
            offs = self._chunk_offs.get((lo, hi))
            if offs is None:
                offs = torch.as_tensor(self._offs_host[lo:hi + 1] - a0, device=self.device)
                self._chunk_offs[(lo, hi)] = offs
        sig = d["sigmas"] if self._shared else d["sigmas"][a0:a1]       # (shared: the molecule's one matrix, the topology stands in for it)
        return d["coords"][a0:a1], offs, sig, d["origins"][lo:hi], None if d["box"] is None else d["box"][lo:hi]

    def voxelize(self, out=None):
        """This rank's shard -> float32 [B_local, V, C] on its device; asynchronous, no collective."""
        return self._run(*self._items(0, self.n_local), out=out)

    def _collectives(self):
        """True when a process group exists: the collectives then run even with ONE rank (a 1-GPU box exercises the
        RCCL path that way); without a group (plain single-process use) everything is local."""
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized()
        except Exception:
            return False

    def gather(self, local, dst=None):
        return gather_features(local, self.bounds, group=self.group, dst=dst) if self._collectives() else local

    def voxelize_gather(self, nchunks=4, dst=None, timings=None, exchange="auto", loopback=False):
        """Voxelize the shard chunk by chunk and gather every finished chunk on a communication stream while the next
        one is computed.  Returns the full float32 [B, V, C] tensor on every rank (``dst=None``: all-gather) or on
        ``dst`` only (others get None).

        ``exchange="p2p"`` (what ``"auto"`` picks for an all-gather, equal or ragged shards): every rank computes
        STRAIGHT INTO its own rows of the result, and each chunk travels as one batch of point-to-point sends / receives
        between the rows' final positions -- no staging tensor, no padding, no second pass over the gathered data, and on
        the xGMI mesh (every GPU wired to every other) all seven links of a GPU carry a chunk at once, where a ring
        all-gather moves it over one link seven times.  ``exchange="allgather"`` (what gather-to-root takes): chunks are
        padded to the largest chunk of any rank so that each step is one equal-sized collective, received into a staging
        tensor and copied to their final rows.  ``loopback``: with ONE rank the point-to-point batch is sent to the rank
        itself and checked (a 1-GPU box exercises the exchange code on RCCL that way); ``self.last_exchange`` names what ran."""
        import torch
        import torch.distributed as dist

        if not self._collectives():
            return self.voxelize()
        ws, rank = self.world, self.rank
        sizes = np.diff(self.bounds)
        if exchange == "auto":
            exchange = "p2p" if dst is None else "allgather"
        if exchange == "p2p" and dst is not None:
            raise ValueError("the point-to-point exchange is an all-gather (dst=None)")
        self.last_exchange = exchange
        cb = [chunk_bounds(int(s), nchunks) for s in sizes]               # per rank: its chunk boundaries (same count)
        tail = (self.V, self.C)
        cuda = self.device.type == "cuda"
        # (a NORMAL-priority stream.  Round 6 tried a high-priority one, as the streamed drivers' copy stream has (batch._stream_voxelize): on
        #  the one-rank RCCL loopback -- the only exchange a one-GPU box can run -- the chunk-overlapped gather cost 10.5 ms beside the compute
        #  instead of 4.6 (two runs each, same box): RCCL's copy kernels then take the CUs ahead of the voxelizer's)
        comm = torch.cuda.Stream(device=self.device) if cuda else None
        main = torch.cuda.current_stream(self.device) if cuda else None

        if exchange == "p2p":
            # every rank computes STRAIGHT INTO its own rows of the result; chunk c of rank r belongs at rows
            # bounds[r] + cb[r][c] .. bounds[r] + cb[r][c+1] on every rank (shards and chunks may be ragged: both sides of a
            # transfer know its size from the partition, an empty chunk is skipped by both)
            full = torch.empty((self.n_items,) + tail, dtype=torch.float32, device=self.device)
            row = lambda r, c: int(self.bounds[r] + cb[r][c])
            # one message holds at most P2P_MAX_BYTES: a 2 GiB send (256 cfg2 grids in ONE chunk) came back different from RCCL's
            # one-rank loopback (round 6, profiles/r6_bench_torchrun1.json) -- a byte count that no longer fits 31 bits.  Both ends
            # of a transfer cut the same rows the same way.
            rows_per_msg = max(1, P2P_MAX_BYTES // max(1, self.V * self.C * 4))

            def pieces(r0, r1):
                return [(a, min(a + rows_per_msg, r1)) for a in range(r0, r1, rows_per_msg)]
            for c in range(len(cb[rank]) - 1):
                lo, hi = int(cb[rank][c]), int(cb[rank][c + 1])
                if hi > lo:
                    self._run(*self._items(lo, hi), out=full[row(rank, c):row(rank, c + 1)])
                if ws == 1 and not loopback:
                    continue
                ev = None
                if cuda:
                    ev = torch.cuda.Event()
                    ev.record(main)
                with (torch.cuda.stream(comm) if cuda else _null()):
                    if cuda:
                        comm.wait_event(ev)
                    ops, check = [], None
                    for k in range(1, ws):                                # staggered peers: rank r starts with r + 1
                        to, frm = (rank + k) % ws, (rank - k) % ws
                        if hi > lo:
                            for a, b in pieces(row(rank, c), row(rank, c + 1)):
                                ops.append(dist.P2POp(dist.isend, full[a:b], self._global_rank(to), group=self.group))
                        if cb[frm][c + 1] > cb[frm][c]:
                            for a, b in pieces(row(frm, c), row(frm, c + 1)):
                                ops.append(dist.P2POp(dist.irecv, full[a:b], self._global_rank(frm), group=self.group))
                    if ws == 1 and hi > lo:
                        # one rank (a 1-GPU box under torchrun): the same batched send / receive, to itself, into a scratch
                        # tensor that must come back equal -- the exchange code touches the communicator at least once
                        check = torch.empty_like(full[row(rank, c):row(rank, c + 1)])
                        ops = []
                        for a, b in pieces(row(rank, c), row(rank, c + 1)):
                            ops.append(dist.P2POp(dist.isend, full[a:b], self._global_rank(rank), group=self.group))
                            ops.append(dist.P2POp(dist.irecv, check[a - row(rank, c):b - row(rank, c)], self._global_rank(rank), group=self.group))
                    if ops:
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()
                    if check is not None and not torch.equal(check, full[row(rank, c):row(rank, c + 1)]):
                        raise RuntimeError("loopback exchange returned different data")
            if cuda:
                main.wait_stream(comm)
                full.record_stream(comm)
            return full

        want = dst is None or rank == dst
        full = torch.empty((self.n_items,) + tail, dtype=torch.float32, device=self.device) if want else None
        local = torch.empty((self.n_local,) + tail, dtype=torch.float32, device=self.device)
        for c in range(len(cb[rank]) - 1):
            lo, hi = int(cb[rank][c]), int(cb[rank][c + 1])
            cmax = max(int(cb[r][c + 1] - cb[r][c]) for r in range(ws))
            if cmax == 0:
                continue
            if hi > lo:
                self._run(*self._items(lo, hi), out=local[lo:hi])
            ev = None
            if cuda:
                ev = torch.cuda.Event()
                ev.record(main)
            with (torch.cuda.stream(comm) if cuda else _null()):
                if cuda:
                    comm.wait_event(ev)
                send = local[lo:hi]
                if hi - lo != cmax:
                    send = torch.zeros((cmax,) + tail, dtype=torch.float32, device=self.device)
                    send[: hi - lo] = local[lo:hi]
                if dst is None:
                    recv = torch.empty((ws * cmax,) + tail, dtype=torch.float32, device=self.device)
                    dist.all_gather_into_tensor(recv, send.contiguous(), group=self.group)
                    parts = [recv[r * cmax:(r + 1) * cmax] for r in range(ws)]
                else:
                    parts = [torch.empty((cmax,) + tail, dtype=torch.float32, device=self.device) for _ in range(ws)] if want else None
                    dist.gather(send.contiguous(), parts, dst=dst, group=self.group)
                if want:
                    for r in range(ws):
                        g0 = int(self.bounds[r] + cb[r][c])
                        n = int(cb[r][c + 1] - cb[r][c])
                        if n:
                            full[g0:g0 + n].copy_(parts[r][:n])
        if cuda:
            main.wait_stream(comm)
            local.record_stream(comm)
        return full

    def _global_rank(self, group_rank):
        """Rank in the default group of rank ``group_rank`` of this voxelizer's group (point-to-point ops address peers
        by their global rank)."""
        import torch.distributed as dist

        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def voxelize_sharded(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None, gather=True,
                     compute=None, device=None, balance_by_atoms=True, nchunks=0, dst=None):
    """One-call form: voxelize a packed batch across all ranks of the default process group.

    Every rank passes the same batch description (host arrays; only views of the rank's own shard are read);
    rank r computes its contiguous shard on its own GPU and -- when ``gather`` -- every rank (or only ``dst``)
    returns the full float32 [B, V, C] tensor, otherwise its local shard.  ``nchunks`` > 0 overlaps the gather
    with the compute chunk by chunk (``ShardedVoxelizer.voxelize_gather``).  Returns (features, bounds).
    """
    sv = ShardedVoxelizer.from_host(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=box,
                                    balance_by_atoms=balance_by_atoms, compute=compute, device=device)
    if not gather or not sv._collectives():
        return sv.voxelize(), sv.bounds
    if nchunks > 0:
        return sv.voxelize_gather(nchunks=nchunks, dst=dst), sv.bounds
    return sv.gather(sv.voxelize(), dst=dst), sv.bounds


# ------------------------------------------------------------------------------------------------
# Frames-sharded distances ("frames shard across GPUs exactly like cfg4").  The reference's frame
# loop is the OUTER loop of every distance_utils function (distance_utils.pyx:149, :244, :316, :76): frames are
# independent, selections / groups are the same for all of them.  So: contiguous frame ranges per rank, the selections
# replicated (a few kilobytes), the trajectory shard resident in HBM in the reference's [N, 3, F_rank] layout, results left
# sharded [F_rank, n_pairs] -- no collective on the compute path -- with the same optional gathers as the voxel path.
# ------------------------------------------------------------------------------------------------
def shard_frames(coords, box, lo, hi):
    """Frames [lo, hi) of a trajectory in the reference's layout: coords [N, 3, F] -> C-contiguous [N, 3, hi - lo], box [3, F] ->
    [3, hi - lo] (frames are the fastest axis, so a frame range is a strided view: this makes the copy a rank uploads)."""
    c = np.ascontiguousarray(np.asarray(coords)[:, :, lo:hi], dtype=np.float32)
    b = None if box is None else np.ascontiguousarray(np.asarray(box)[:, lo:hi], dtype=np.float32)
    return c, b


class ShardedDistances:
    """This rank's frames of a trajectory, resident on its device, and the distance_utils functions over them.

    ``from_host(coords [N,3,F], box [3,F])`` (every rank passes the same arrays; only this rank's frames are copied) or
    ``from_loader(n_frames, loader)`` with ``loader(lo, hi) -> (coords [N,3,hi-lo], box [3,hi-lo] | None)`` (no rank holds the
    whole trajectory: e.g. a rank decodes its own frames of an XTC file).  ``weights`` (per frame) balance ragged costs.
    Every method returns this rank's rows ``[F_rank, n_pairs]`` as a float32 tensor on its device, asynchronously, with NO
    collective; ``gather(local)`` / ``gather(local, dst=0)`` collect the rows of all ranks in frame order (one padded RCCL
    all-gather / gather-to-root over xGMI, as for the features).  The kernels are those of ``moleculekit_amd.distance_utils``
    (same bits).  ``compute(kind, coords, box, *args) -> ndarray``: CPU test-suite only (the sharding / gather logic under
    ``gloo``); the product never injects it.
    """

    def __init__(self, n_frames, bounds, shard, device=None, compute=None, group=None, ctx=None):
        import torch

        self.group = group
        self.rank, self.world = world(group)
        self.n_frames = int(n_frames)
        self.bounds = np.asarray(bounds, dtype=np.int64)
        if len(self.bounds) != self.world + 1 or self.bounds[0] != 0 or self.bounds[-1] != self.n_frames:
            raise ValueError("bounds must partition range(n_frames) over the ranks")
        self.lo, self.hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        coords, box = shard
        coords = np.asarray(coords)
        if coords.ndim != 3 or coords.shape[1] != 3 or coords.shape[2] != self.hi - self.lo:
            raise ValueError(f"this rank's shard must be coords [N, 3, {self.hi - self.lo}], got {coords.shape}")
        self.n_atoms, self.n_local = int(coords.shape[0]), int(coords.shape[2])
        if box is None:                                              # (what the reference's drivers pass for periodic=None)
            box = np.zeros((3, self.n_local), dtype=np.float32)
        if np.asarray(box).shape != (3, self.n_local):
            raise ValueError(f"box must be [3, {self.n_local}] for this rank's frames")
        self._compute = compute
        self._ctx = ctx
        if compute is None:
            if device is None:
                device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", self.rank)) % max(torch.cuda.device_count(), 1))
            self.device = torch.device(device)
            if self.device.type != "cuda":
                raise RuntimeError("ShardedDistances needs a HIP device (there is no CPU path); tests inject `compute`")
            # pinned staging -> one H2D copy per array (the trajectory shard is the only big thing a rank ever uploads)
            self._coords = self._to_device(np.ascontiguousarray(coords, dtype=np.float32))
            self._box = self._to_device(np.ascontiguousarray(box, dtype=np.float32))
        else:
            self.device = torch.device("cpu" if device is None else device)
            self._coords = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float32))
            self._box = torch.from_numpy(np.ascontiguousarray(box, dtype=np.float32))
        self._cache = {}

    # ---- construction -------------------------------------------------------------------------------------
    @classmethod
    def from_host(cls, coords, box=None, weights=None, **kw):
        rank, ws = world(kw.get("group"))
        F = int(np.asarray(coords).shape[2])
        bounds = shard_bounds(F, ws, weights)
        return cls(F, bounds, shard_frames(coords, box, int(bounds[rank]), int(bounds[rank + 1])), **kw)

    @classmethod
    def from_loader(cls, n_frames, loader, weights=None, **kw):
        rank, ws = world(kw.get("group"))
        bounds = shard_bounds(int(n_frames), ws, weights)
        return cls(n_frames, bounds, tuple(loader(int(bounds[rank]), int(bounds[rank + 1]))), **kw)

    def _to_device(self, a):
        import torch
        pinned = torch.empty(a.shape, dtype=getattr(torch, a.dtype.name), pin_memory=True)
        np.copyto(pinned.numpy(), a)
        t = pinned.to(self.device, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()          # (set-up: the staging buffer is released here)
        return t

    def _dev(self, key, a, dtype):
        """Replicated small inputs (selections, groups, chain ids, masses): uploaded once per distinct array."""
        import torch
        a = np.ascontiguousarray(a, dtype=dtype)
        k = (key, a.shape, a.tobytes() if a.nbytes <= 1 << 16 else hash(a.tobytes()))
        t = self._cache.get(k)
        if t is None:
            # (uint32 has no torch dtype everywhere: the bit patterns travel as int32 / int64)
            view = a.view(np.int32) if a.dtype == np.uint32 else a
            t = torch.as_tensor(view, device=self.device)
            if len(self._cache) > 64:
                self._cache.clear()
            self._cache[k] = t
        return t

    def _context(self):
        from . import _lib
        ctx = self._ctx or _lib.default_context(self.device.index if self.device.index is not None else 0)
        return ctx

    # ---- the distance_utils functions over this rank's frames ---------------------------------------------------------
    def dist_trajectory(self, sel1, sel2, digitized_chains, selfdist, pbc, out=None, squared=False):
        """distance_utils.pyx:126-155 over frames [lo, hi): float32 [F_rank, n_pairs]."""
        import torch
        sel1, sel2 = np.ascontiguousarray(sel1, np.uint32), np.ascontiguousarray(sel2, np.uint32)
        chains = np.ascontiguousarray(digitized_chains, np.uint32)
        n1, n2 = len(sel1), len(sel2)
        P = n1 * n2 if not selfdist else sum(max(n2 - 1 - i, 0) for i in range(n1))
        if self._compute is not None:
            res = torch.from_numpy(np.asarray(self._compute("dist_trajectory", self._coords.numpy(), self._box.numpy(), sel1, sel2, chains,
                                                           bool(selfdist), bool(pbc)), dtype=np.float32).reshape(self.n_local, P))
            return res if out is None else out.copy_(res)
        for name, s in (("sel1", sel1), ("sel2", sel2)):
            if len(s) and int(s.max()) >= self.n_atoms:
                raise ValueError(f"{name} index out of range")
        if len(chains) < self.n_atoms:
            raise ValueError(f"digitized_chains has {len(chains)} entries for {self.n_atoms} atoms")
        if out is None:
            out = torch.empty((self.n_local, P), dtype=torch.float32, device=self.device)
        if self.n_local and P:
            ctx = self._context()
            ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
            try:
                ctx.dist_trajectory_dev(self._coords, self.n_local, self._box, self._dev("s1", sel1, np.uint32), n1, self._dev("s2", sel2, np.uint32), n2,
                                        self._dev("ch", chains, np.uint32), bool(selfdist), bool(pbc), bool(squared), out)
            finally:
                ctx.set_stream(None)
        return out

    def dist_trajectory_reduction(self, groups1, groups2, digitized_chains1, digitized_chains2, selfdist, pbc, masses, reduction1, reduction2,
                                  pairs=False, out=None):
        """distance_utils.pyx:211-350 (``pairs``: the ``_pairs`` form) over frames [lo, hi): float32 [F_rank, n_group_pairs]."""
        import torch
        from .distance_utils import _csr
        a1, o1 = _csr(groups1); a2, o2 = _csr(groups2)
        ng1, ng2 = len(groups1), len(groups2)
        if pairs and ng1 != ng2:
            raise ValueError("pairs mode needs the same number of groups on both sides")
        P = ng1 if pairs else (ng1 * ng2 if not selfdist else sum(max(ng2 - 1 - i, 0) for i in range(ng1)))
        ch1, ch2 = np.ascontiguousarray(digitized_chains1, np.uint32), np.ascontiguousarray(digitized_chains2, np.uint32)
        masses = np.ascontiguousarray(masses, np.float32)
        if self._compute is not None:
            res = torch.from_numpy(np.asarray(self._compute("dist_trajectory_reduction", self._coords.numpy(), self._box.numpy(), groups1, groups2, ch1, ch2,
                                                           bool(selfdist), bool(pbc), masses, int(reduction1), int(reduction2), bool(pairs)),
                                              dtype=np.float32).reshape(self.n_local, P))
            return res if out is None else out.copy_(res)
        if (a1.size and (a1.min() < 0 or a1.max() >= self.n_atoms)) or (a2.size and (a2.min() < 0 or a2.max() >= self.n_atoms)):
            raise ValueError("group atom index out of range")
        if min((len(g) for g in groups1), default=1) == 0 or min((len(g) for g in groups2), default=1) == 0:
            raise ValueError("empty group")
        if len(ch1) < ng1 or len(ch2) < ng2 or len(masses) < self.n_atoms:
            raise ValueError("digitized_chains1/2 need one entry per group, masses one per atom")
        if out is None:
            out = torch.empty((self.n_local, P), dtype=torch.float32, device=self.device)
        if self.n_local and P:
            ctx = self._context()
            ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
            try:
                ctx.dist_reduction_dev(self._coords, self.n_atoms, self.n_local, self._box, self._dev("a1", a1, np.int32), self._dev("o1", o1, np.int64), ng1,
                                       len(a1), self._dev("a2", a2, np.int32), self._dev("o2", o2, np.int64), ng2, self._dev("c1", ch1, np.uint32),
                                       self._dev("c2", ch2, np.uint32), bool(selfdist), bool(pairs), bool(pbc), self._dev("m", masses, np.float32),
                                       int(reduction1), int(reduction2), out)
            finally:
                ctx.set_stream(None)
        return out

    def contacts_trajectory(self, sel1, sel2, digitized_chains, selfdist, pbc, dist_threshold=5):
        """distance_utils.pyx:59-93 over frames [lo, hi): ``(frame_offsets int64 [F_rank + 1] (host), pairs uint32 [n, 2] tensor on the
        device)`` -- frame f of this rank owns rows ``frame_offsets[f] : frame_offsets[f + 1]``; thresholded and compacted on the GPU."""
        import torch
        sel1, sel2 = np.ascontiguousarray(sel1, np.uint32), np.ascontiguousarray(sel2, np.uint32)
        chains = np.ascontiguousarray(digitized_chains, np.uint32)
        if self._compute is not None:
            offs, pairs = self._compute("contacts_trajectory", self._coords.numpy(), self._box.numpy(), sel1, sel2, chains, bool(selfdist), bool(pbc),
                                        float(dist_threshold))
            return np.asarray(offs, np.int64), torch.from_numpy(np.asarray(pairs, np.int64).reshape(-1, 2))
        if not self.n_local:
            return np.zeros(1, np.int64), torch.zeros((0, 2), dtype=torch.int64, device=self.device)
        ctx = self._context()
        ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        try:
            offs, ptr, n = ctx.contacts_trajectory_dev(self._coords, self.n_local, self._box, self._dev("s1", sel1, np.uint32), len(sel1),
                                                       self._dev("s2", sel2, np.uint32), len(sel2), self._dev("ch", chains, np.uint32),
                                                       bool(selfdist), bool(pbc), float(dist_threshold))
            pairs = torch.empty((n, 2), dtype=torch.int32, device=self.device)
            if n:                                   # the list is context-owned until the next contacts call: a device-to-device copy keeps it
                from . import _lib
                _lib._check(_lib.load().mkamd_copy_dev(ctx._h, pairs.data_ptr(), ptr, n * 8))
        finally:
            ctx.set_stream(None)
        return offs, pairs

    # ---- the optional collectives (after the compute) ----------------------------------------------------------------
    def _collectives(self):
        return ShardedVoxelizer._collectives(self)

    def gather(self, local, dst=None):
        """Rows of every rank in frame order: [F, n_pairs] on every rank (``dst=None``: one padded all-gather) or on ``dst`` only."""
        return gather_features(local, self.bounds, group=self.group, dst=dst) if self._collectives() else local
