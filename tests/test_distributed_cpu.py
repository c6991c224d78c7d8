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
    # which lines of voxelize_gather this rank executes (the world-8 run has to ENTER every branch the 8-GPU line takes)
    code = D.ShardedVoxelizer.voxelize_gather.__code__
    lines = set()

    def tracer(frame, event, arg):
        if frame.f_code is not code:
            return None

        def local(fr, ev, a):
            if ev == "line":
                lines.add(fr.f_lineno)
            return local
        return local

    sys.settrace(tracer)
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
            # (round 6: a point-to-point message is capped -- RCCL returned a 2 GiB loopback message changed; with a cap of one byte
            #  every row travels as a message of its own, the same rows cut the same way at both ends)
            D.P2P_MAX_BYTES = 1 if B == 19 else 1 << 30
            lo, hi = int(sv.bounds[rank]), int(sv.bounds[rank + 1])
            if B == 5:
                ok = ok and int((np.diff(sv.bounds) == 0).sum()) == 3
            for nchunks in (1, 4):
                got = sv.voxelize_gather(nchunks=nchunks, exchange="p2p")
                ok = ok and sv.last_exchange == "p2p" and np.array_equal(got.numpy(), ref)
                got = sv.voxelize_gather(nchunks=nchunks, exchange="allgather")
                ok = ok and np.array_equal(got.numpy(), ref)
                got = sv.voxelize_gather(nchunks=nchunks, dst=0)                # gather-to-root (padded collectives)
                ok = ok and ((got is None) if rank else np.array_equal(got.numpy(), ref))
            ok = ok and np.array_equal(sv.voxelize().numpy(), ref[lo:hi])
            detail.append([int(b) for b in sv.bounds])
        sys.settrace(None)
        q.put((rank, bool(ok), detail, sorted(lines)))
    finally:
        sys.settrace(None)
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
    assert all(r[1] for r in res), [r[:3] for r in res]
    for b in res[0][2]:
        assert b[0] == 0 and len(b) == world + 1
    # branch coverage of voxelize_gather over the eight ranks: every executable line was entered by some rank, except the ones that
    # cannot run in this setting BY THEIR OWN TEXT (no process group, one rank's loopback, the CUDA-only stream plumbing, the argument check)
    import inspect
    from moleculekit_amd import distributed as D
    fn = D.ShardedVoxelizer.voxelize_gather
    src, first = inspect.getsourcelines(fn)
    executable = {ln for _, _, ln in fn.__code__.co_lines() if ln is not None and ln > first}
    hit = set().union(*[set(r[3]) for r in res])
    allowed = ("return self.voxelize()", "ws == 1", "loopback", "check", "raise ValueError", "if cuda", "comm.wait_event", "torch.cuda.Event()", "ev.record(main)",
               "main.wait_stream(comm)", "record_stream(comm)", "import torch")
    missing = {ln: src[ln - first].strip() for ln in sorted(executable - hit)}

    def explained(ln):
        """the line itself, or the statement that controls it (the nearest line above with a smaller indentation), says why"""
        i = ln - first
        if any(a in src[i] for a in allowed):
            return True
        ind = len(src[i]) - len(src[i].lstrip())
        for j in range(i - 1, 0, -1):                       # every statement that controls it, innermost first
            t = src[j]
            if t.strip() and len(t) - len(t.lstrip()) < ind:
                if any(a in t for a in allowed):
                    return True
                ind = len(t) - len(t.lstrip())
        return False

    unexplained = {ln: t for ln, t in missing.items() if not explained(ln)}
    assert not unexplained, unexplained
    assert len(hit) > 40, len(hit)


# ------------------------------------------------------------------------------------------------
# frames-sharded distances (moleculekit_amd.distributed.ShardedDistances): the oracle stands in for the
# kernels, what is under test is the frame partition (ragged, empty shards), the replicated selections, results left sharded
# [F_rank, n_pairs] and the two gathers.
# ------------------------------------------------------------------------------------------------
def _dist_compute(kind, coords, box, *a):
    from oracle import oracle
    if kind == "dist_trajectory":
        sel1, sel2, chains, selfdist, pbc = a
        return oracle.dist_trajectory(coords, box, sel1, sel2, chains, selfdist, pbc)
    if kind == "dist_trajectory_reduction":
        g1, g2, c1, c2, selfdist, pbc, masses, r1, r2, pairs = a
        return oracle.dist_trajectory_reduction(coords, box, g1, g2, c1, c2, selfdist, pbc, masses, r1, r2, pairs=pairs)
    sel1, sel2, chains, selfdist, pbc, thr = a
    d2 = oracle.dist_trajectory(coords, box, sel1, sel2, chains, selfdist, pbc, squared=True)
    ii, jj = (np.triu_indices(len(sel1), 1) if selfdist else [x.ravel() for x in np.indices((len(sel1), len(sel2)))])
    offs, rows = [0], []
    for f in range(d2.shape[0]):
        hit = np.nonzero(d2[f] <= np.float32(thr) * np.float32(thr))[0]
        rows.append(np.stack([sel1[ii[hit]], sel2[jj[hit]]], 1).astype(np.int64))
        offs.append(offs[-1] + len(hit))
    return np.asarray(offs, np.int64), (np.concatenate(rows) if rows else np.zeros((0, 2), np.int64))


def _worker_dist(rank, world, port, F, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from moleculekit_amd import distributed as D
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(31)
        N = 40
        coords = rng.uniform(-20, 20, size=(N, 3, F)).astype(np.float32)
        box = rng.uniform(25, 33, size=(3, F)).astype(np.float32)
        chains = rng.integers(0, 3, N).astype(np.uint32)
        masses = rng.uniform(1, 16, N).astype(np.float32)
        sel1 = np.sort(rng.choice(N, 7, replace=False)).astype(np.uint32)
        sel2 = np.sort(rng.choice(N, 9, replace=False)).astype(np.uint32)
        g1 = [rng.choice(N, int(rng.integers(1, 6)), replace=False).tolist() for _ in range(5)]
        g2 = [rng.choice(N, int(rng.integers(1, 6)), replace=False).tolist() for _ in range(4)]
        ch1, ch2 = rng.integers(0, 2, 5).astype(np.uint32), rng.integers(0, 2, 4).astype(np.uint32)
        seen = []

        def loader(lo, hi):                                    # a rank that only ever sees its own frames
            seen.append((lo, hi))
            return D.shard_frames(coords, box, lo, hi)

        ok = True
        for sd in (D.ShardedDistances.from_host(coords, box, compute=_dist_compute),
                   D.ShardedDistances.from_loader(F, loader, compute=_dist_compute)):
            lo, hi = sd.lo, sd.hi
            ok = ok and (lo, hi) == (int(sd.bounds[rank]), int(sd.bounds[rank + 1])) and sd.n_local == hi - lo
            # dist_trajectory: rows of this rank, then both gathers in frame order
            want = oracle.dist_trajectory(coords, box, sel1, sel2, chains, False, True)
            local = sd.dist_trajectory(sel1, sel2, chains, False, True)
            ok = ok and tuple(local.shape) == (hi - lo, want.shape[1]) and np.array_equal(local.numpy(), want[lo:hi])
            ok = ok and np.array_equal(sd.gather(local).numpy(), want)
            root = sd.gather(local, dst=0)
            ok = ok and ((root is None) if rank else np.array_equal(root.numpy(), want))
            want = oracle.dist_trajectory(coords, box, sel2, sel2, chains, True, False)
            ok = ok and np.array_equal(sd.gather(sd.dist_trajectory(sel2, sel2, chains, True, False)).numpy(), want)
            # group reductions (closest and centre of mass, all-vs-all and pairs)
            want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, True, masses, 0, 0)
            ok = ok and np.array_equal(sd.gather(sd.dist_trajectory_reduction(g1, g2, ch1, ch2, False, True, masses, 0, 0)).numpy(), want)
            want = oracle.dist_trajectory_reduction(coords, box, g1[:4], g2, ch1[:4], ch2, False, True, masses, 1, 0, pairs=True)
            ok = ok and np.array_equal(sd.gather(sd.dist_trajectory_reduction(g1[:4], g2, ch1[:4], ch2, False, True, masses, 1, 0, pairs=True)).numpy(), want)
            # contact lists: per-rank offsets and rows; stitched together over the ranks they are the whole trajectory's lists
            offs, pairs = sd.contacts_trajectory(sel1, sel2, chains, False, True, 14.0)
            woffs, wpairs = _dist_compute("contacts_trajectory", coords, box, sel1, sel2, chains, False, True, 14.0)
            ok = ok and len(offs) == hi - lo + 1 and np.array_equal(np.diff(offs), np.diff(woffs[lo:hi + 1]))
            ok = ok and np.array_equal(pairs.numpy(), wpairs[woffs[lo]:woffs[hi]])
        ok = ok and seen == [(int(sd.bounds[rank]), int(sd.bounds[rank + 1]))]
        # per-frame weights balance a ragged cost (e.g. frames of different system sizes do not exist here; the partition follows them anyway)
        w = np.ones(F); w[: F // 2] = 3.0
        sdw = D.ShardedDistances.from_host(coords, box, weights=w, compute=_dist_compute)
        want = oracle.dist_trajectory(coords, box, sel1, sel2, chains, False, False)
        ok = ok and np.array_equal(sdw.gather(sdw.dist_trajectory(sel1, sel2, chains, False, False)).numpy(), want)
        q.put((rank, bool(ok), [int(b) for b in sd.bounds]))
    finally:
        dist.destroy_process_group()


def _run_world_dist(world, F):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dist, args=(r, world, port, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    b = res[0][2]
    assert b[0] == 0 and b[-1] == F and len(b) == world + 1


@pytest.mark.parametrize("F", [9, 2, 1])
def test_sharded_distances_world2_gloo(F):
    """Frames-sharded distance_utils under gloo, two ranks: 9 frames (5 + 4), 2 (1 + 1), 1 (an empty shard)."""
    _run_world_dist(2, F)


def test_sharded_distances_world8_gloo():
    """Eight ranks, 19 frames (shards of 2-3 frames) -- the 8-GPU node's shape -- and the padded collectives over ragged shards."""
    _run_world_dist(8, 19)


def test_sharded_distances_single_process_without_a_group():
    """No process group: everything is local, gather() is the identity."""
    from moleculekit_amd import distributed as D
    from oracle import oracle
    rng = np.random.default_rng(3)
    coords = rng.uniform(-10, 10, size=(12, 3, 5)).astype(np.float32)
    sd = D.ShardedDistances.from_host(coords, None, compute=_dist_compute)
    sel = np.arange(4, dtype=np.uint32)
    got = sd.dist_trajectory(sel, sel + 4, np.zeros(12, np.uint32), False, False)
    assert np.array_equal(sd.gather(got).numpy(), oracle.dist_trajectory(coords, np.zeros((3, 5), np.float32), sel, sel + 4, np.zeros(12, np.uint32), False, False))
    with pytest.raises(ValueError):
        D.ShardedDistances(5, [0, 4], (coords, None), compute=_dist_compute)


def test_shared_sigmas_is_validated_with_an_injected_compute_too():
    """`shared_sigmas=True` used to be ignored silently when `compute` was injected.  Now: the molecule's [n, C] matrix is
    accepted and repeated for the stand-in, a repeated matrix has to BE a repeat, ragged items are refused."""
    import torch
    from moleculekit_amd import distributed as D
    n, F = 5, 3
    rng = np.random.default_rng(0)
    coords = rng.normal(size=(n * F, 3)).astype(np.float32)
    offs = np.arange(F + 1) * n
    sig1 = rng.uniform(1, 2, size=(n, 8))
    origins = np.zeros((F, 3))
    seen = {}

    def compute(c, o, s, org, nv, vs, bx):
        seen["sig"] = s.copy()
        return torch.zeros((len(o) - 1, int(np.prod(nv)), s.shape[1]))

    for sig in (sig1, np.tile(sig1, (F, 1))):
        sv = D.ShardedVoxelizer.from_host(coords, offs, sig, origins, [2, 2, 2], 1.0, compute=compute, shared_sigmas=True)
        sv.voxelize()
        assert seen["sig"].shape == (n * F, 8) and np.array_equal(seen["sig"], np.tile(sig1, (F, 1)))
    bad = np.tile(sig1, (F, 1)); bad[-1, 0] += 1
    with pytest.raises(ValueError):
        D.ShardedVoxelizer.from_host(coords, offs, bad, origins, [2, 2, 2], 1.0, compute=compute, shared_sigmas=True)
    with pytest.raises(ValueError):
        D.ShardedVoxelizer.from_host(coords, np.array([0, 4, 10, 15]), np.tile(sig1, (F, 1)), origins, [2, 2, 2], 1.0, compute=compute, shared_sigmas=True)
