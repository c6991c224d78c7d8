"""CPU tier: the N>1 path (item sharding + the trivial gather) under torch.distributed/gloo with
world_size 2.  The per-rank compute is injected (the oracle stands in for the HIP kernels, which need
a GPU); what is under test is moleculekit_amd.distributed: partition, shard slicing, padded all-gather
and gather-to-root, ragged shards."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from moleculekit_amd import distributed as D
    from tests.cases import oracle_lattice
    from tests.synth import grid_origin, synth_config

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = synth_config(5, B)                    # ragged molecules (20..50 atoms)
        origins = np.stack([grid_origin(c, p["boxsize"], 1.0)[0] for c in p["centers"]])
        nv = np.array([10, 9, 8])

        def compute(c, offs, s, o, nvox, vs, bx):
            return torch.from_numpy(oracle_lattice(c, offs, s, o, nvox, vs, bx).astype(np.float32))

        full, bounds = D.voxelize_sharded(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0,
                                          gather=True, compute=compute)
        local, _ = D.voxelize_sharded(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0,
                                      gather=False, compute=compute)
        root = D.gather_features(local, bounds, dst=0)
        ref = oracle_lattice(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0).astype(np.float32)
        ok = (tuple(full.shape) == ref.shape and np.array_equal(full.numpy(), ref)
              and local.shape[0] == bounds[rank + 1] - bounds[rank]
              and np.array_equal(local.numpy(), ref[bounds[rank]:bounds[rank + 1]])
              and ((root is None) if rank != 0 else np.array_equal(root.numpy(), ref)))
        # the chunk-overlapped gather (all ranks / root only), on ragged shards balanced by atoms, more chunks than items
        for nchunks in (1, 3, 16):
            ch, b2 = D.voxelize_sharded(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0, gather=True,
                                        compute=compute, nchunks=nchunks)
            ok = ok and np.array_equal(b2, bounds) and np.array_equal(ch.numpy(), ref)
            ch0, _ = D.voxelize_sharded(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0, gather=True,
                                        compute=compute, nchunks=nchunks, dst=0)
            ok = ok and ((ch0 is None) if rank != 0 else np.array_equal(ch0.numpy(), ref))
        # a rank that only ever sees its own shard (loader), partition balanced by atoms per item
        seen = []

        def loader(lo, hi):
            seen.append((lo, hi))
            return D.shard_packed(p["coords"], p["atom_offsets"], p["sigmas"], origins, None, lo, hi)

        sv = D.ShardedVoxelizer.from_loader(B, loader, nv, 1.0, weights=np.diff(p["atom_offsets"]) + 1.0, compute=compute)
        ok = ok and seen == [(int(sv.bounds[rank]), int(sv.bounds[rank + 1]))] and np.array_equal(sv.bounds, bounds)
        ok = ok and np.array_equal(sv.voxelize_gather(nchunks=2).numpy(), ref)
        # equal shards (plain partition of an even batch): computed straight into the result, chunks exchanged point to
        # point; the same values as the staged all-gather, whatever the chunking
        if B % world == 0:
            eq = D.ShardedVoxelizer.from_host(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0,
                                              balance_by_atoms=False, compute=compute)
            ok = ok and np.all(np.diff(eq.bounds) == B // world)
            for nchunks in (1, 2, 5):
                ok = ok and np.array_equal(eq.voxelize_gather(nchunks=nchunks).numpy(), ref)
                ok = ok and np.array_equal(eq.voxelize_gather(nchunks=nchunks, exchange="allgather").numpy(), ref)
            ok = ok and np.array_equal(eq.gather(eq.voxelize()).numpy(), ref)
        # ragged shards (balanced by atoms): since round 4 the point-to-point exchange takes them too (both ends of a transfer
        # know its size from the partition; an empty chunk is skipped by both) -- it is what "auto" picks for an all-gather
        for nchunks in (1, 3):
            got = sv.voxelize_gather(nchunks=nchunks, exchange="p2p")
            ok = ok and sv.last_exchange == "p2p" and np.array_equal(got.numpy(), ref)
            ok = ok and np.array_equal(sv.voxelize_gather(nchunks=nchunks, exchange="allgather").numpy(), ref) and sv.last_exchange == "allgather"
        try:
            sv.voxelize_gather(nchunks=2, exchange="p2p", dst=0)          # the point-to-point exchange is an all-gather
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok), [int(b) for b in bounds]))
    finally:
        dist.destroy_process_group()


def _run_world(world, B):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    b = res[0][2]
    assert b[0] == 0 and b[-1] == B and len(b) == world + 1


@pytest.mark.parametrize("B", [7, 8, 2, 1])
def test_sharded_voxelization_world2_gloo(B):
    _run_world(2, B)


@pytest.mark.parametrize("B", [11, 3])
def test_sharded_voxelization_world4_gloo(B):
    """Four ranks (round 4): every rank exchanges with THREE peers per chunk -- the batched isend / irecv pattern of the
    8-GPU run, where gloo runs with two ranks proved little; 11 ragged molecules over 4 ranks (shards of 2-4 items, chunks
    of 0-2), and 3 items over 4 ranks (an empty shard: a rank that only receives)."""
    _run_world(4, B)


def _worker8(rank, world, port, q):
    """world 8: the point-to-point exchange with SEVEN peers per chunk, ragged shards (balanced by atoms) and empty ones."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist

    from moleculekit_amd import distributed as D
    from tests.synth import grid_origin, synth_config

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        detail = []
        for B in (19, 5):                        # 19 ragged molecules over 8 ranks (shards of 1-4); 5 over 8 (three empty shards)
            p = synth_config(5, B)
            origins = np.stack([grid_origin(c, p["boxsize"], 1.0)[0] for c in p["centers"]])
            nv = np.array([6, 5, 4])

            def compute(c, offs, s, o, nvox, vs, bx):
                # a function of the item alone (its atom count and coordinate sum): a misplaced row shows
                n = len(offs) - 1
                csum = np.concatenate([[0.0], np.cumsum(c.astype(np.float64).sum(axis=1))])
                val = 1.0 + np.diff(offs) + 1e-3 * (csum[offs[1:]] - csum[offs[:-1]])
                return torch.from_numpy(np.broadcast_to(val[:, None, None], (n, int(np.prod(nvox)), s.shape[1])).astype(np.float32).copy())

            ref = compute(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0, None).numpy()
            sv = D.ShardedVoxelizer.from_host(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0,
                                              balance_by_atoms=(B == 19), compute=compute)
            lo, hi = int(sv.bounds[rank]), int(sv.bounds[rank + 1])
            if B == 5:
                ok = ok and int((np.diff(sv.bounds) == 0).sum()) == 3
            for nchunks in (1, 4):
                got = sv.voxelize_gather(nchunks=nchunks, exchange="p2p")
                ok = ok and sv.last_exchange == "p2p" and np.array_equal(got.numpy(), ref)
                got = sv.voxelize_gather(nchunks=nchunks, exchange="allgather")
                ok = ok and np.array_equal(got.numpy(), ref)
            ok = ok and np.array_equal(sv.voxelize().numpy(), ref[lo:hi])
            detail.append([int(b) for b in sv.bounds])
        q.put((rank, bool(ok), detail))
    finally:
        dist.destroy_process_group()


def test_sharded_voxelization_world8_gloo():
    """Eight ranks: `voxelize_gather(exchange="p2p", nchunks in {1, 4})` with ragged and with empty
    shards -- every rank posts up to seven sends and seven receives per chunk, the shape of the 8-GPU node's exchange."""
    import torch.multiprocessing as mp

    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    for b in res[0][2]:
        assert b[0] == 0 and len(b) == world + 1
