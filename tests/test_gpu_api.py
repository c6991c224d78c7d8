"""GPU tier (-m gpu): the reference-shaped Python API on top of the C ABI -- written to read like the
reference's own tests (tests/test_voxeldescriptors.py): compute, load the stored reference result,
np.allclose / np.array_equal."""
import os

import numpy as np
import pytest

from tests.cases import TOL, golden
from tests.synth import grid_origin, synth_config
from tests.test_host_logic import Mol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["celecoxib", "ledipasvir"])
def test_small_molecule_like_reference_test(name):
    """test_voxeldescriptors.py:41-68 (channel 7 = occupancies; channels 0-6 need RDKit typing)."""
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = golden(f"{name}_ch7.npz")
    mol = Mol(g["coords"], element=g["element"])
    ch = np.zeros((len(g["coords"]), 8), dtype=bool)
    ch[:, 7] = g["element"] != "H"
    features, centers, nvoxels = getVoxelDescriptors(mol, buffer=1, userchannels=ch)
    assert np.allclose(features[:, 7], g["ref_features_ch7"])
    assert np.abs(features[:, 7] - g["ref_features_ch7"]).max() <= TOL
    assert np.array_equal(centers, g["ref_centers"])
    assert np.array_equal(nvoxels, g["ref_nvoxels"])
    assert not features[:, :7].any()


def test_cfg1_getvoxeldescriptors_boxsize_branch():
    """BASELINE.json configs[0]: 3PTB, 24^3 @ 1 A, 8 channels."""
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = golden("cfg1_3ptb.npz")
    for method in ("C", "hip"):
        features, centers, nvoxels = getVoxelDescriptors(
            None, boxsize=[24, 24, 24], center=g["center"], voxelsize=1,
            usercoords=g["coords"], userchannels=g["sigmas"], method=method)
        assert features.dtype == np.float64 and features.flags["C_CONTIGUOUS"]
        assert np.allclose(features, g["features"])
        assert np.abs(features - g["features"]).max() <= TOL
        assert np.array_equal(centers, g["centers"]) and np.array_equal(nvoxels, g["nvoxels"])


def test_usercenters_paths():
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = golden("cfg1_3ptb.npz")
    # explicit, non-lattice centres -> (features, centers) and the same `centers` object comes back
    uc = np.array([[0.0, 0, 0], [16, 24, -5], [10.2, 3.3, 25.1]]) + g["center"] * [1, 0, 0]
    from oracle import oracle
    feats, cen = getVoxelDescriptors(None, usercenters=uc, usercoords=g["coords"], userchannels=g["sigmas"])
    assert cen is uc and feats.shape == (3, 8)
    assert np.abs(feats - oracle.calculate_occupancy(uc, g["coords"], g["sigmas"])).max() <= TOL
    # lattice handed in as usercenters is recognised and takes the tiled kernel; same numbers
    feats2, cen2 = getVoxelDescriptors(None, usercenters=g["centers"], usercoords=g["coords"][:, :, None],
                                       userchannels=g["sigmas"])
    assert np.abs(feats2 - g["features"]).max() <= TOL


def test_bool_channels_use_vdw_radii():
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = golden("3ptb_bbox_buffer8.npz")
    mol = Mol(g["coords"], element=g["element"])
    fb, _, _ = getVoxelDescriptors(mol, boxsize=[16, 16, 16], center=g["coords"].mean(0), userchannels=g["userchannels"])
    ff, _, _ = getVoxelDescriptors(mol, boxsize=[16, 16, 16], center=g["coords"].mean(0),
                                   userchannels=g["radii"][:, None] * g["userchannels"].astype(float))
    assert np.array_equal(fb, ff)


def test_trajectory_and_batch_wrappers():
    from moleculekit_amd import batch
    from tests.cases import case_cfg4_small, check
    case = case_cfg4_small()
    F = len(case["atom_offsets"]) - 1
    n = case["atom_offsets"][1]
    xyz = np.transpose(case["coords"].reshape(F, n, 3), (1, 2, 0))           # Molecule.coords layout [N,3,F]
    box = case["box"].T                                                       # Molecule.box layout [3,F]
    feats, origin, nvox = batch.voxelizeTrajectory(xyz, case["sigmas"][:n], center=case["origins"][0] + 12.0,
                                                   boxsize=[24, 24, 24], voxelsize=1, box=box)
    check(case, feats)
    g = golden("cfg3_small.npz")
    B = int(g["nmol"])
    cl = [g["coords"][g["atom_offsets"][b]:g["atom_offsets"][b + 1]] for b in range(B)]
    sl = [g["sigmas"][g["atom_offsets"][b]:g["atom_offsets"][b + 1]] for b in range(B)]
    feats, origins, nvox = batch.getVoxelDescriptorsBatch(cl, sl, g["centers"], g["boxsize"], float(g["voxelsize"]))
    assert np.abs(feats - g["features"]).max() <= TOL and np.array_equal(nvox, g["nvoxels"])


def test_torch_device_resident_path_matches_host_path():
    import torch
    from moleculekit_amd import batch
    from tests.cases import case_ragged_batch, case_pbc_batch, check
    for case in (case_ragged_batch(), case_pbc_batch()):
        dev = torch.device("cuda", 0)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
        mi = 1 if case["box"] is None else batch.max_images_per_atom(case["box"], case["nvoxels"], case["voxelsize"])
        out = batch.voxelize_lattice_torch(
            t(case["coords"], np.float32), t(case["atom_offsets"], np.int64), t(case["sigmas"], np.float32),
            t(case["origins"], np.float64), case["nvoxels"], case["voxelsize"],
            box=None if case["box"] is None else t(case["box"], np.float32), max_images=mi)
        torch.cuda.synchronize()
        check(case, out.cpu().numpy(), tol=2e-5 if case["sigmas"].dtype != np.float32 else 1e-5)
        cf = batch.voxelize_lattice_torch(
            t(case["coords"], np.float32), t(case["atom_offsets"], np.int64), t(case["sigmas"], np.float32),
            t(case["origins"], np.float64), case["nvoxels"], case["voxelsize"],
            box=None if case["box"] is None else t(case["box"], np.float32), max_images=mi, channel_first=True)
        nx, ny, nz = [int(v) for v in case["nvoxels"]]
        assert cf.shape == (out.shape[0], out.shape[2], nx, ny, nz)
        # no copy was made: the [B,V,C] buffer IS the channels-last-3d (NDHWC) layout of the logical [B,C,nx,ny,nz] tensor
        assert cf.is_contiguous(memory_format=torch.channels_last_3d)
        assert torch.equal(cf.permute(0, 2, 3, 4, 1).reshape(out.shape), out)


def test_fused_rotation_augmentation_on_device():
    """per-item rotateCoordinates (voxeldescriptors.py:78-114) fused into the device pipeline + the
    channel-first view the reference's tutorial builds by hand before nn.Conv3d."""
    import torch
    from moleculekit_amd import batch
    from moleculekit_amd.voxeldescriptors import rotateCoordinates
    from tests.cases import LATTICE_CASES, oracle_lattice
    case = LATTICE_CASES["cfg3_small"]()
    B = len(case["atom_offsets"]) - 1
    rng = np.random.default_rng(43)
    rots = rng.uniform(-np.pi, np.pi, size=(B, 3))
    cens = case["origins"] + 12.0
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    fused = batch.voxelize_lattice_torch(t(case["coords"], np.float32), t(case["atom_offsets"], np.int64),
                                         t(case["sigmas"], np.float64), t(case["origins"], np.float64), case["nvoxels"],
                                         case["voxelsize"], affine=t(batch.rotation_affines(rots, cens), np.float64),
                                         channel_first=True)
    torch.cuda.synchronize()
    nx, ny, nz = [int(v) for v in case["nvoxels"]]
    assert fused.shape == (B, 8, nx, ny, nz)
    got = fused.permute(0, 2, 3, 4, 1).reshape(B, -1, 8).cpu().numpy()
    rc = case["coords"].copy()
    for b in range(B):
        s, e = case["atom_offsets"][b], case["atom_offsets"][b + 1]
        rc[s:e] = rotateCoordinates(case["coords"][s:e], list(rots[b]), cens[b]).astype(np.float32)
    exp = oracle_lattice(rc, case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    assert np.abs(got - exp).max() <= TOL
    assert np.abs(got - case["expected"]).max() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("version", [1, 2])
def test_getvoxeldescriptors_types_channels_itself_for_pdbqt_molecules(version):
    """No `userchannels`: the channels come from the molecule's PDBQT atom types (moleculekit_amd.channels; the
    real moleculekit's getChannels would be used if it were importable).  Expected = oracle on the sigma
    channels the REFERENCE's typing produced for the same molecule (tests/golden/channels_3ptb.npz)."""
    import copy

    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "channels_3ptb.npz"))
    mol = Mol(g["coords"], element=g["element"])
    mol.atomtype, mol.name, mol.resname, mol.charge, mol.bonds = g["atomtype"], g["name"], g["resname"], g["charge"], g["bonds"]
    mol.copy = lambda: copy.copy(mol)
    center = g["coords"].mean(0).astype(np.float64)
    feats, centers, nvox = getVoxelDescriptors(mol, boxsize=[16, 16, 16], center=center, voxelsize=1, version=version)
    sig = g["channels_v1"] if version == 1 else g["radii"][:, None] * g["feats_v2"].astype(float)
    from oracle import oracle
    want = oracle.calculate_occupancy(centers, g["coords"], sig)
    assert feats.shape == want.shape == (16 ** 3, 8)
    assert np.abs(feats - want).max() <= 1e-5


@pytest.mark.gpu
def test_1atl_metalloprotein_from_its_typed_molecule():
    """The fixture pair the reference's own test holds (tests/test_voxeldescriptors.py:109-131: 1ATL_atomtyped ->
    1ATL_channels.npy, zinc and calcium in the metal channel): getVoxelDescriptors types the molecule itself
    (version 2, validity checks on) and voxelizes it around the zinc site; expected = oracle on the channel matrix
    the REFERENCE stored for that molecule."""
    import copy

    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "atomtyper_1atl.npz"))
    mol = Mol(g["coords"], element=g["element"])
    for f in ("atomtype", "name", "resname", "resid", "insertion", "chain", "segid", "bonds", "bondtype", "charge"):
        setattr(mol, f, g[f])
    mol.copy = lambda: copy.copy(mol)
    zn = g["coords"][g["element"] == "Zn"][0].astype(np.float64)
    feats, centers, nvox = getVoxelDescriptors(mol, boxsize=[24, 24, 24], center=zn, voxelsize=1, version=2)
    from oracle import oracle
    want = oracle.calculate_occupancy(centers, g["coords"], g["ref_channels"])
    assert feats.shape == want.shape == (24 ** 3, 8) and list(nvox) == [24, 24, 24]
    assert np.abs(feats - want).max() <= TOL
    assert feats[:, 6].max() > 0.99 and (feats[:, 6] > 0).sum() < 1500       # the metal channel: two spheres, nothing else


def test_streamed_trajectory_matches_the_all_at_once_call():
    """SURVEY 8f-4 (trajectory feeding): chunks uploaded on a copy stream while the previous chunk is voxelized;
    same values as voxelizeTrajectory, whatever the chunking, with and without a periodic box / frame subsets."""
    import torch
    from moleculekit_amd import batch
    rng = np.random.default_rng(9)
    N, F = 300, 37
    L = 26.0
    xyz = rng.uniform(0, L, size=(N, 3, F)).astype(np.float32)
    sig = np.where(rng.random((N, 8)) < 0.4, rng.choice([1.2, 1.7, 1.55], size=(N, 1)), 0.0)
    box = np.full((3, F), L, np.float32)
    for bx, frames in ((None, None), (box, None), (box, np.array([3, 4, 5, 20, 7, 36]))):
        want, origin, nv = batch._voxelizeTrajectory_packed(xyz, sig, [L / 2] * 3, [16, 16, 16], 1.0, box=bx, frames=frames)
        pub, origin2, nv2 = batch.voxelizeTrajectory(xyz, sig, [L / 2] * 3, [16, 16, 16], 1.0, box=bx, frames=frames)   # (streamed inside)
        assert np.array_equal(pub, want) and np.array_equal(origin2, origin) and np.array_equal(nv2, nv) and pub.flags["C_CONTIGUOUS"]
        for chunk in (5, 16, 64):
            seen, outs = [], []
            for idx, feats in batch.iterVoxelizeTrajectory(xyz, sig, [L / 2] * 3, [16, 16, 16], 1.0, box=bx, frames=frames, chunk=chunk):
                assert feats.is_cuda and feats.shape[1:] == (16 ** 3, 8)
                seen.append(np.asarray(idx)); outs.append(feats)
            torch.cuda.synchronize()
            got = torch.cat(outs).cpu().numpy()
            assert np.array_equal(np.concatenate(seen), np.arange(F) if frames is None else frames)
            assert np.array_equal(got, want)
    cf = next(batch.iterVoxelizeTrajectory(xyz, sig, [L / 2] * 3, [16, 16, 16], 1.0, chunk=4, channel_first=True))[1]
    assert cf.shape == (4, 8, 16, 16, 16)


def test_xtc_fed_stream_matches_decode_then_voxelize():
    """XTC file -> host-thread decode straight into pinned staging -> copy stream -> voxelizer, against
    XTCread (host decode + unit conversion) followed by voxelizeTrajectory; periodic (the file's box) and not."""
    import torch
    from moleculekit_amd import batch, xtc
    fn = os.path.join(os.path.dirname(__file__), "golden", "xtc", "aladipep.xtc")
    tr = xtc.XTCread(fn)
    N, F = tr.coords.shape[0], tr.coords.shape[2]
    rng = np.random.default_rng(12)
    sig = np.where(rng.random((N, 8)) < 0.4, rng.choice([1.1, 1.52, 1.7], size=(N, 1)), 0.0)
    center = tr.coords[:, :, 0].mean(0).astype(np.float64)
    for pbc in (False, True):
        want, _, _ = batch._voxelizeTrajectory_packed(tr.coords, sig, center, [14, 14, 14], 1.0, box=tr.box.astype(np.float32) if pbc else None)
        for chunk in (7, 64):
            for decode in ("auto", "gpu", "host"):                # (auto = the device decoder for this contiguous range)
                seen, outs = [], []
                for idx, f in batch.iterVoxelizeXTC(fn, sig, center, [14, 14, 14], 1.0, pbc=pbc, chunk=chunk, decode=decode):
                    seen.append(np.asarray(idx)); outs.append(f)
                torch.cuda.synchronize()
                assert np.array_equal(np.concatenate(seen), np.arange(F))
                assert np.array_equal(torch.cat(outs).cpu().numpy(), want), (pbc, chunk, decode)
    sel = np.array([5, 0, 19])
    want, _, _ = batch._voxelizeTrajectory_packed(tr.coords, sig, center, [14, 14, 14], 1.0, frames=sel)
    for decode in ("auto", "gpu", "host"):                        # (auto = the host decoder for a scattered selection)
        outs = [f for _, f in batch.iterVoxelizeXTC(fn, sig, center, [14, 14, 14], 1.0, pbc=False, frames=sel, chunk=2, decode=decode)]
        assert np.array_equal(torch.cat(outs).cpu().numpy(), want), decode
    with pytest.raises(ValueError, match="decode"):
        next(batch.iterVoxelizeXTC(fn, sig, center, [14, 14, 14], 1.0, decode="fpga"))


def test_streaming_paths_raise_on_bad_frames_instead_of_yielding_incomplete_features():
    """A trajectory whose box shrinks after the first frame (NPT) keeps its periodic images: every chunk recomputes the
    images per atom from ITS boxes (same values as the all-at-once call, which sees all boxes); a frame with a box edge
    <= 10 A -- or a zero box -- raises instead of silently dropping atoms; and what only the device can see (more images
    than a caller of the raw torch entry point reserved) comes back through poll_errors / synchronize."""
    import torch
    from moleculekit_amd import _lib, batch
    rng = np.random.default_rng(19)
    N, F = 200, 24
    xyz = rng.uniform(0, 30, size=(N, 3, F)).astype(np.float32)
    sig = np.where(rng.random((N, 8)) < 0.4, rng.choice([1.2, 1.7], size=(N, 1)), 0.0)
    box = np.full((3, F), 40.0, np.float32)
    box[:, 8:] = 13.0                                         # smaller than grid + halo: several images per atom from frame 8 on
    want, _, _ = batch._voxelizeTrajectory_packed(xyz, sig, [15.0] * 3, [20, 20, 20], 1.0, box=box)
    assert np.array_equal(batch.voxelizeTrajectory(xyz, sig, [15.0] * 3, [20, 20, 20], 1.0, box=box)[0], want)
    outs = [f for _, f in batch.iterVoxelizeTrajectory(xyz, sig, [15.0] * 3, [20, 20, 20], 1.0, box=box, chunk=4)]
    torch.cuda.synchronize()
    assert np.array_equal(torch.cat(outs).cpu().numpy(), want)
    bad = box.copy(); bad[1, 13] = 9.5
    with pytest.raises(ValueError):
        for _ in batch.iterVoxelizeTrajectory(xyz, sig, [15.0] * 3, [20, 20, 20], 1.0, box=bad, chunk=4):
            pass
    # the raw device entry point with too few reserved images: the flag is raised on the device, poll_errors reports it
    ctx = _lib.default_context(0)
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    fr = np.ascontiguousarray(np.transpose(xyz[:, :, 8:10], (2, 0, 1))).reshape(-1, 3)
    out = batch.voxelize_lattice_torch(t(fr, np.float32), t(np.arange(3) * N, np.int64), t(np.tile(sig, (2, 1)), np.float32),
                                       t(np.tile([[5.0, 5.0, 5.0]], (2, 1)), np.float64), [20, 20, 20], 1.0,
                                       box=t(np.full((2, 3), 13.0), np.float32), max_images=1, ctx=ctx)
    torch.cuda.synchronize()
    with pytest.raises(_lib.MkamdError):
        ctx.poll_errors()
    ctx.poll_errors()                                         # reported once, then clean again
    del out


def test_host_call_in_two_halves(hip_ctx):
    """mkamd_voxelize_lattice_host_begin / _end (the drop-in getVoxelDescriptors copies its voxel centres between the two):
    the same bits as the one-piece call, float32 and float64; an `end` without a `begin` is an error; a `begin` that is
    never ended is abandoned by the next one."""
    from moleculekit_amd import batch, _lib
    g = golden("cfg1_3ptb.npz")
    from tests.synth import grid_origin
    o, nv = grid_origin(g["center"], g["boxsize"], float(g["voxelsize"]))
    args = (g["coords"], np.array([0, len(g["coords"])]), g["sigmas"], o[None], nv, float(g["voxelsize"]))
    ref = batch.voxelize_lattice(*args, ctx=hip_ctx)
    end = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)
    busy = np.arange(100000).sum()                                         # (host work beside the device)
    got = end()
    assert got.dtype == np.float32 and np.array_equal(got, ref) and busy > 0
    got64 = batch.voxelize_lattice_begin(*args, ctx=hip_ctx, dtype=np.float64)()
    assert got64.dtype == np.float64 and np.array_equal(got64, ref.astype(np.float64))
    with pytest.raises(ValueError, match="no host call was begun"):
        hip_ctx.voxelize_lattice_host_end(np.empty_like(ref))
    batch.voxelize_lattice_begin(*args, ctx=hip_ctx)                       # never ended ...
    again = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)()             # ... the next call is not disturbed
    assert np.array_equal(again, ref)
    assert np.abs(ref[0].astype(np.float64) - g["features"]).max() <= 1e-5


def test_usercenters_guess_is_checked(hip_ctx):
    """Centres brought by the caller: the kernels start on the lattice recognised in the LAST array of that length while
    this one is checked.  Right guess (same grid, and the same grid shifted), wrong guess (another voxel size with the same
    number of centres; the same centres jittered: no lattice at all) -- every call equals what the explicit path gives."""
    from moleculekit_amd import voxeldescriptors as vd
    from moleculekit_amd.voxeldescriptors import getCenters, getVoxelDescriptors
    g = golden("cfg1_3ptb.npz")
    coords, chans = g["coords"], g["sigmas"]
    a, _ = getCenters(boxsize=[24, 24, 24], center=g["center"], voxelsize=1)
    b, _ = getCenters(boxsize=[12, 12, 12], center=g["center"], voxelsize=0.5)          # also 24^3 centres
    assert a.shape == b.shape
    rng = np.random.default_rng(1)
    jit = a + rng.normal(0, 1e-3, a.shape)
    def direct(c):                                                                      # boxsize / explicit path, no guessing
        vd._LAST_LATTICE.clear()
        return getVoxelDescriptors(None, usercenters=c, usercoords=coords, userchannels=chans)[0]
    ref = {k: direct(c) for k, c in (("a", a), ("a+", a + 0.25), ("b", b), ("jit", jit))}
    assert np.abs(ref["a"] - g["features"]).max() <= TOL
    vd._LAST_LATTICE.clear()
    for k, c in (("a", a), ("a", a), ("a+", a + 0.25), ("b", b), ("b", b), ("jit", jit), ("jit", jit), ("a", a), ("a", a)):
        got = getVoxelDescriptors(None, usercenters=c, usercoords=coords, userchannels=chans)[0]
        assert np.array_equal(got, ref[k]), k


@pytest.mark.gpu
def test_promised_calls_and_the_streaming_drivers_are_pipelined_and_bitwise_equal(hip_ctx):
    """Round 4: pipelining is a PRODUCT path.  (1) the per-call promise (mkamd_ctx_promise_inputs), with the inputs complete
    already and with inputs produced on ANOTHER stream that come with an event; (2) ShardedVoxelizer.voxelize promises its
    resident shard; (3) iterVoxelizeTrajectory prepares chunk k+1 on its copy stream and hands the library the event --
    host source and device-resident source.  Every one of them bitwise equal to the in-order run, and the context's
    counter says the pre-passes really went to the side stream."""
    import torch
    from moleculekit_amd import batch
    from moleculekit_amd.distributed import ShardedVoxelizer
    dev = torch.device("cuda", hip_ctx.device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    hip_ctx.set_pipelining(False)
    # (1) the raw promise
    p = synth_config(2, 5, seed=71)
    o = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    w = (t(p["coords"], np.float32), t(p["atom_offsets"], np.int64), t(p["sigmas"], np.float32), t(o, np.float64), nv, p["voxelsize"])
    ref = [batch.voxelize_lattice_torch(*w, ctx=hip_ctx).cpu().numpy() for _ in range(2)]
    n0 = hip_ctx.pipelined_calls()
    outs = []
    for _ in range(4):
        hip_ctx.promise_inputs(None)
        outs.append(batch.voxelize_lattice_torch(*w, ctx=hip_ctx))
    torch.cuda.synchronize(); hip_ctx.synchronize()
    assert hip_ctx.pipelined_calls() - n0 == 4
    assert all(np.array_equal(x.cpu().numpy(), ref[0]) for x in outs)
    # inputs produced on another stream: a scaled copy made there, the event handed over, nothing awaited by the caller
    side = torch.cuda.Stream(device=dev)
    outs = []
    for i in range(3):
        with torch.cuda.stream(side):
            for _ in range(20):                                   # (something in front of it, so that the copy really is late)
                junk = w[0] * 1.0001
            c2 = (w[0] * 0.5).mul_(2.0)                           # == w[0] bit for bit
            ev = torch.cuda.Event(); ev.record(side)
        hip_ctx.promise_inputs(ev)
        outs.append((batch.voxelize_lattice_torch(c2, *w[1:], ctx=hip_ctx), c2, junk))
    torch.cuda.synchronize(); hip_ctx.synchronize()
    assert all(np.array_equal(x[0].cpu().numpy(), ref[0]) for x in outs)
    hip_ctx.promise_inputs(None); hip_ctx.withdraw_promise()      # a withdrawn promise pipelines nothing
    n1 = hip_ctx.pipelined_calls()
    assert np.array_equal(batch.voxelize_lattice_torch(*w, ctx=hip_ctx).cpu().numpy(), ref[0]) and hip_ctx.pipelined_calls() == n1
    # (2) the sharded voxelizer over its resident shard
    got = {}
    for piped in (False, True):
        sv = ShardedVoxelizer.from_host(p["coords"], p["atom_offsets"], p["sigmas"].astype(np.float32), o, nv, p["voxelsize"], device=dev, ctx=hip_ctx,
                                        pipelined=piped)
        n1 = hip_ctx.pipelined_calls()
        outs = [sv.voxelize() for _ in range(3)]
        torch.cuda.synchronize(); hip_ctx.synchronize()
        assert (hip_ctx.pipelined_calls() - n1 == 3) == piped
        got[piped] = [x.cpu().numpy() for x in outs]
    assert all(np.array_equal(a, ref[0]) for a in got[False] + got[True])
    # (3) the streamed trajectory: 12 frames of 30 000 atoms in chunks of 8 (240 000 atoms per call: a big call)
    q = synth_config(4, 12, seed=72)
    N = 30000
    xyz = np.ascontiguousarray(q["coords"].reshape(12, N, 3).transpose(1, 2, 0))      # [N, 3, F]
    sig = q["sigmas"][:N]
    box = np.ascontiguousarray(q["box"].T)                                             # [3, F]
    center = q["centers"][0]
    res = {}
    for name, src, bx in (("host", xyz, box), ("device", t(xyz, np.float32), box)):
        for piped in (False, True):
            n1 = hip_ctx.pipelined_calls()
            outs = [f for _, f in batch.iterVoxelizeTrajectory(src, sig, center, [32, 32, 32], 1.0, box=bx, chunk=8, ctx=hip_ctx,
                                                               pipelined=piped)]
            torch.cuda.synchronize()
            assert (hip_ctx.pipelined_calls() - n1 >= 1) == piped, (name, piped)
            res[name, piped] = torch.cat(outs).cpu().numpy()
    want, _, _ = batch._voxelizeTrajectory_packed(xyz, sig, center, [32, 32, 32], 1.0, box=box)
    for k, v in res.items():
        assert np.array_equal(v, want), k
    assert want.max() > 0.5


def test_host_call_halves_are_guarded(hip_ctx):
    """Round 4 (ADVICE): `end` writes through a bare pointer -- the array must be float32 / float64, C-contiguous and hold exactly
    the pending call's B*V*C values (the library checks the count itself); and between `begin` and `end` the context refuses
    every entry point that would regrow or overwrite what `end` hands back, while queries and the centre generator pass."""
    from moleculekit_amd import batch
    g = golden("cfg1_3ptb.npz")
    o, nv = grid_origin(g["center"], g["boxsize"], float(g["voxelsize"]))
    args = (g["coords"], np.array([0, len(g["coords"])]), g["sigmas"], o[None], nv, float(g["voxelsize"]))
    ref = batch.voxelize_lattice(*args, ctx=hip_ctx)
    with pytest.raises(ValueError, match="float32 or float64"):
        batch.voxelize_lattice_begin(*args, ctx=hip_ctx, dtype=np.float16)
    held = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)                                 # (kept: a dropped second half abandons the call, below)
    with pytest.raises(ValueError):
        hip_ctx.voxelize_lattice_host_end(np.empty(ref.size, dtype=np.int32))           # not a float array
    with pytest.raises(ValueError, match="does not hold"):
        hip_ctx.voxelize_lattice_host_end(np.empty(ref.size - 8, dtype=np.float32))      # too small: refused by the library
    with pytest.raises(ValueError, match="no host call was begun"):                      # ... and the call was abandoned
        hip_ctx.voxelize_lattice_host_end(np.empty_like(ref))
    end = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)
    assert np.array_equal(batch.voxelize_lattice(*args, ctx=hip_ctx), ref)               # a host call is its own begin: the pending one is
    with pytest.raises(ValueError, match="no host call was begun"):                      # abandoned (documented), its `end` says so
        end()
    end = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)
    import torch
    dev = torch.device("cuda", hip_ctx.device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    with pytest.raises(ValueError, match="has not been ended"):                          # the device entry point shares the workspace
        batch.voxelize_lattice_torch(t(g["coords"], np.float32), t(args[1], np.int64), t(g["sigmas"], np.float32), t(o[None], np.float64),
                                     nv, float(g["voxelsize"]), ctx=hip_ctx)
    with pytest.raises(ValueError, match="has not been ended"):
        batch.occupancy_centers(np.zeros((4, 3)), g["coords"], g["sigmas"], ctx=hip_ctx)
    cen = batch.grid_centers(o, nv, float(g["voxelsize"]), ctx=hip_ctx)                  # allowed: touches neither
    hip_ctx.poll_errors()
    assert np.array_equal(end(), ref) and cen.shape == (int(np.prod(nv)), 3)
    assert np.array_equal(batch.voxelize_lattice(*args, ctx=hip_ctx), ref)
    del held
    # a second half that is DROPPED (an exception between the halves) must not leave the shared context refusing
    # everything: the object's finalizer gives the call up (mkamd_ctx_abandon_pending) ...
    import gc
    end = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)
    with pytest.raises(ValueError, match="has not been ended"):
        batch.occupancy_centers(np.zeros((4, 3)), g["coords"], g["sigmas"], ctx=hip_ctx)
    del end
    gc.collect()
    assert batch.occupancy_centers(np.zeros((4, 3)), g["coords"], g["sigmas"], ctx=hip_ctx).shape == (4, g["sigmas"].shape[1])
    # ... but only its OWN call: an old object collected after a newer begin leaves the newer call alone
    old = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)
    new = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)                               # (abandons `old`'s call in the library)
    del old
    gc.collect()
    assert np.array_equal(new(), ref)
    # and explicitly, from C callers' point of view: abandon, then every entry point again
    end = batch.voxelize_lattice_begin(*args, ctx=hip_ctx)
    end.abandon()
    with pytest.raises(ValueError, match="no host call was begun"):
        hip_ctx.voxelize_lattice_host_end(np.empty_like(ref))
    assert np.array_equal(batch.voxelize_lattice(*args, ctx=hip_ctx), ref)
    # the drop-in call whose centre generation fails between the halves leaves the context usable
    from moleculekit_amd import voxeldescriptors as vd
    real = vd._centersFromSpec
    vd._centersFromSpec = lambda *a, **k: (_ for _ in ()).throw(MemoryError("no room for the centres"))
    try:
        with pytest.raises(MemoryError):
            vd.getVoxelDescriptors(None, boxsize=[24, 24, 24], center=list(g["center"]), usercoords=g["coords"], userchannels=g["sigmas"])
    finally:
        vd._centersFromSpec = real
    gc.collect()
    assert batch.occupancy_centers(np.zeros((4, 3)), g["coords"], g["sigmas"], ctx=hip_ctx).shape[0] == 4


def test_device_xtc_decoder_is_bit_exact_with_the_host_decoder(hip_ctx, tmp_path):
    """Round 4 (csrc/xtc_gpu.h): a GPU lane decodes a frame.  Every reference-held trajectory of the test tier (runs of small
    differences, the water swap, step changes of the small-number table), a synthetic 30 000-atom file, per-axis bit fields,
    <= 9 atoms as plain floats, frame subsets in any order -- coordinates bit for bit what the host decoder (itself pinned
    against the REAL reference reader, tests/test_xtc.py) returns, scaled to Angstrom with the same float32 multiply; box
    vectors, time and step the same; a range beyond 64 bits is refused, not mis-decoded."""
    import torch
    from moleculekit_amd import xtc
    here = os.path.join(os.path.dirname(__file__), "golden", "xtc")
    files = [os.path.join(here, n + ".xtc") for n in ("mol", "aladipep", "3ptb_traj_head", "4rws_head")]
    rng = np.random.default_rng(77)
    for name, N, F, L in (("syn", 30000, 5, 6.69), ("small", 7, 4, 2.0), ("wide", 50, 3, 30000.0), ("tiny_box", 200, 6, 0.8)):
        x = (rng.uniform(-0.3, 1.0, size=(N, 3, F)) * L).astype(np.float32)
        bv = np.zeros((3, 3, F), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = L
        fn = str(tmp_path / (name + ".xtc"))
        xtc.write_xtc(fn, x, bv, np.arange(F, dtype=np.float32), np.arange(F))
        files.append(fn)
    for fn in files:
        F = xtc.get_xtc_nframes(fn)
        for sel in (None, np.arange(F)[::-1][: max(1, F // 2)], np.array([F - 1, 0, F - 1])):
            c, b, t, s = xtc.read_xtc(fn) if sel is None else xtc.read_xtc_frames(fn, sel)
            for scale in (1.0, 10.0):
                xyz, b2, t2, s2 = xtc.read_xtc_frames_dev(fn, sel, scale=scale, ctx=hip_ctx)
                want = np.ascontiguousarray(np.transpose(c, (2, 0, 1))) * np.float32(scale)
                got = xyz.cpu().numpy()
                assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (os.path.basename(fn), scale)
                assert np.array_equal(b2, b) and np.array_equal(t2, t) and np.array_equal(s2, s)
    x = (rng.uniform(-0.5, 1.0, size=(40, 3, 2)) * 3000.0).astype(np.float32)            # 66..72-bit mixed-radix numbers
    bv = np.zeros((3, 3, 2), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = 3000.0
    fn = str(tmp_path / "huge.xtc")
    xtc.write_xtc(fn, x, bv, np.zeros(2, np.float32), np.arange(2))
    with pytest.raises(RuntimeError, match="more than 64 bits"):
        xtc.read_xtc_frames_dev(fn, ctx=hip_ctx)
    assert not xtc.device_decodable(xtc.chunk_desc(fn, np.arange(2), 40)[0], 40)


@pytest.mark.parametrize("name", ["mol", "aladipep", "3ptb_traj_head", "4rws_head"])
def test_device_xtc_decoder_against_the_reference_readers_output(hip_ctx, name):
    """VERDICT r4, parity thin spot (a): the device decoder (csrc/xtc_gpu.h) compared DIRECTLY with what the real reference
    reader (fileformats/xtc/src/xdrfile.cpp:749-983, through tests/golden/make_golden_xtc.py) decoded from the same
    reference-held files -- not only with this package's host decoder: every stored atom bit for bit, the integer sum of ALL
    coordinate bit patterns, box vectors, times, steps, and the golden's frame selection."""
    from moleculekit_amd import xtc
    here = os.path.join(os.path.dirname(__file__), "golden", "xtc")
    g = np.load(os.path.join(here, name + "_decoded.npz"))
    fn = os.path.join(here, name + ".xtc")
    st = int(g["stride"])
    xyz, box, time, step = xtc.read_xtc_frames_dev(fn, None, scale=1.0, ctx=hip_ctx)          # [F, N, 3] on the device, nm
    got = np.ascontiguousarray(np.transpose(xyz.cpu().numpy(), (1, 2, 0)))                    # -> the reader's [N, 3, F]
    assert got.dtype == np.float32 and got.shape[0] == int(g["natoms"])
    assert np.array_equal(got[::st].view(np.uint32), g["coords"].view(np.uint32))
    assert int(got.view(np.uint32).astype(np.uint64).sum()) == int(g["bitsum"])               # every atom, not just the stored ones
    assert np.array_equal(box, g["box"]) and np.array_equal(time, g["time"]) and np.array_equal(step, g["step"])
    xyz, box, time, step = xtc.read_xtc_frames_dev(fn, g["sel"], scale=1.0, ctx=hip_ctx)
    got = np.ascontiguousarray(np.transpose(xyz.cpu().numpy(), (1, 2, 0)))
    assert np.array_equal(got[::st].view(np.uint32), g["sel_coords"].view(np.uint32)) and np.array_equal(box, g["sel_box"])
    assert np.array_equal(time, g["sel_time"]) and np.array_equal(step, g["sel_step"])


def test_device_xtc_decoder_many_waves_many_windows_and_bad_streams(hip_ctx, tmp_path):
    """The two passes of csrc/xtc_gpu.h at sizes where their bookkeeping matters: several hundred frames (a wave walks 64,
    lanes finish at different positions, the last wave is ragged), streams of tens of LDS windows per frame, chunks of
    frames that differ in length; the streamed form (``iterVoxelizeXTC(decode="gpu")``: growing byte buffers, two slots,
    statuses checked as chunks complete) against the host decoder's; a damaged stream raises -- from the one-off reader and
    from the stream -- instead of yielding coordinates."""
    import torch
    from moleculekit_amd import batch, xtc
    src = os.path.join(os.path.dirname(__file__), "golden", "xtc", "3ptb_traj_head.xtc")
    blob = open(src, "rb").read()
    many = str(tmp_path / "many.xtc")
    with open(many, "wb") as fh:
        for _ in range(35):
            fh.write(blob)
    rng = np.random.default_rng(5)
    N, F = 3000, 150
    walk = np.cumsum(rng.normal(0, 0.02, size=(N, 3, 1)), axis=0) + rng.normal(0, 0.01, size=(N, 3, F)) + rng.uniform(0, 4.0, size=(1, 3, F))
    bv = np.zeros((3, 3, F), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = 5.0
    syn = str(tmp_path / "walk.xtc")
    xtc.write_xtc(syn, walk.astype(np.float32), bv, np.arange(F, dtype=np.float32), np.arange(F))
    for fn in (many, syn):
        c, b, t, s = xtc.read_xtc(fn)
        xyz, b2, t2, s2 = xtc.read_xtc_frames_dev(fn, scale=10.0, ctx=hip_ctx)
        want = np.ascontiguousarray(np.transpose(c, (2, 0, 1))) * np.float32(10.0)
        assert np.array_equal(xyz.cpu().numpy().view(np.uint32), want.view(np.uint32)), os.path.basename(fn)
        assert np.array_equal(b2, b) and np.array_equal(t2, t) and np.array_equal(s2, s)
    # streamed: device decode against host decode, chunk sizes that leave ragged tails and make the byte buffers grow
    na, nf = xtc.get_xtc_natoms(many), xtc.get_xtc_nframes(many)
    sig = np.where(rng.random((na, 4)) < 0.3, 1.6, 0.0)
    center = xtc.read_xtc_frames(many, np.array([0]))[0][:, :, 0].mean(0).astype(np.float64) * 10.0
    want = torch.cat([f for _, f in batch.iterVoxelizeXTC(many, sig, center, [12, 12, 12], 1.0, pbc=True, chunk=64, decode="host")]).cpu().numpy()
    for chunk in (50, 128):
        got = torch.cat([f for _, f in batch.iterVoxelizeXTC(many, sig, center, [12, 12, 12], 1.0, pbc=True, chunk=chunk, decode="gpu")]).cpu().numpy()
        assert got.shape[0] == nf and np.array_equal(got, want), chunk
    # (round 6) chunks approached through a ramp -- 8, 16, 32, 64, 64, ... frames: the plan's sizes, every frame once and in order, the same features
    for decode in ("gpu", "host"):
        parts = list(batch.iterVoxelizeXTC(many, sig, center, [12, 12, 12], 1.0, pbc=True, chunk=64, ramp=8, decode=decode))
        assert [len(i) for i, _ in parts] == list(np.diff(batch.chunk_plan(nf, 64, 8))) and [len(i) for i, _ in parts][:4] == [8, 16, 32, 64]
        assert np.array_equal(np.concatenate([i for i, _ in parts]), np.arange(nf))
        assert np.array_equal(torch.cat([f for _, f in parts]).cpu().numpy(), want), decode
    # a damaged stream: an impossible small-number index in one frame's header (xdrfile.cpp:782 reads it before the stream)
    desc = xtc.chunk_desc(many, np.arange(3), na)[0].view(xtc.DESC_DTYPE).reshape(-1)
    bad = bytearray(open(many, "rb").read())
    # the record's smallidx word sits 8 bytes before the stream's byte count, which is 4 bytes before the stream
    off = int(desc["data_off"][1]) + xtc.chunk_desc(many, np.arange(3), na)[1] - 8
    assert int.from_bytes(bad[off:off + 4], "big") == int(desc["smallidx"][1])
    bad[off:off + 4] = (200).to_bytes(4, "big")
    badfn = str(tmp_path / "bad.xtc")
    open(badfn, "wb").write(bytes(bad))
    with pytest.raises((RuntimeError, ValueError)):
        xtc.read_xtc_frames(badfn, np.arange(3))
    with pytest.raises(RuntimeError, match="corrupt"):
        xtc.read_xtc_frames_dev(badfn, np.arange(3), ctx=hip_ctx)
    with pytest.raises(RuntimeError, match="corrupt"):
        for _ in batch.iterVoxelizeXTC(badfn, sig, center, [12, 12, 12], 1.0, pbc=False, frames=np.arange(3), chunk=2, decode="gpu"):
            pass


def test_cfg4_full_count_streamed_from_xtc_device_decode_equals_host_decode(hip_ctx, tmp_path):
    """BASELINE.json's cfg4 at its full count through the feeder: 10 000 frames of 30 000 atoms in a periodic
    66.9 A box, read from an XTC file and voxelized onto the 48^3 x 8 grid chunk by chunk -- once with the coordinates
    decompressed on the device (csrc/xtc_gpu.h) and once by the host decoder (pinned against the real reference reader in
    tests/test_xtc.py).  3.5 TB of features are not kept: every chunk is reduced on the device to two checksums (the sum of
    its bit patterns as integers -- any differing bit of any element shows -- and the float64 sum), and the frame indices,
    the chunk sizes and both checksums of every chunk must agree; a sample of frames is also compared element by element."""
    import torch
    from moleculekit_amd import batch, xtc
    from tools.benchlib import workloads as bench            # (the workload generators; bench.py itself is not needed)
    base = 128
    p, _, _ = bench.make_workload("cfg4", base, seed=4004)
    N = int(p["atom_offsets"][1])
    L = float(p["box"][0, 0])
    nm = np.ascontiguousarray((p["coords"].reshape(base, N, 3) * np.float32(0.1)).transpose(1, 2, 0))
    bv = np.zeros((3, 3, base), np.float32)
    bv[0, 0] = bv[1, 1] = bv[2, 2] = L * 0.1
    one = str(tmp_path / "one.xtc")
    xtc.write_xtc(one, nm, bv, np.arange(base, dtype=np.float32), np.arange(base))
    blob = open(one, "rb").read()
    per = len(blob) // base
    fn = str(tmp_path / "cfg4_full.xtc")
    with open(fn, "wb") as fh:
        for _ in range(10000 // base):
            fh.write(blob)
        fh.write(blob[:(10000 % base) * per])                      # (frames are self-contained records of equal size here)
    assert xtc.get_xtc_nframes(fn) == 10000 and xtc.get_xtc_natoms(fn) == N
    sig = np.ascontiguousarray(p["sigmas"][:N], dtype=np.float32)

    def run(decode):
        sums, keep = [], {}
        for idx, feats in batch.iterVoxelizeXTC(fn, sig, p["centers"][0], p["boxsize"], p["voxelsize"], pbc=True, chunk=1000, ctx=hip_ctx, decode=decode):
            assert feats.shape == (len(idx), 48 ** 3, 8)
            sums.append((int(idx[0]), len(idx), int(feats.view(torch.int32).sum(dtype=torch.int64)), float(feats.sum(dtype=torch.float64))))
            for f in (0, 4999, 9999):
                if idx[0] <= f <= idx[-1]:
                    keep[f] = feats[f - int(idx[0])].cpu().numpy()
            del feats
        return sums, keep

    dev_sums, dev_keep = run("gpu")
    host_sums, host_keep = run("host")
    assert len(dev_sums) == 10 and sum(s[1] for s in dev_sums) == 10000
    assert dev_sums == host_sums
    assert all(np.array_equal(dev_keep[f], host_keep[f]) for f in (0, 4999, 9999)) and dev_keep[0].max() > 0.5
    # frame 9 999 is a copy of base frame 9 999 % 128 = 15 (frames are self-contained): decoded from another place of the stream,
    # voxelized in another chunk -- the same features as that frame on its own (at the tile depth the big calls take: a small
    # call's K = 4 rounds the plane sweep differently, ~3e-6; csrc/pipeline.h)
    hip_ctx.set_tile_k(8)
    try:
        alone = next(batch.iterVoxelizeXTC(fn, sig, p["centers"][0], p["boxsize"], p["voxelsize"], pbc=True, frames=np.array([9999 % base]), ctx=hip_ctx))[1]
    finally:
        hip_ctx.set_tile_k(0)
    assert np.array_equal(alone[0].cpu().numpy(), dev_keep[9999])


def test_topology_calls_are_bitwise_the_plain_calls_and_refuse_what_they_cannot_serve(hip_ctx):
    """Round 5: frames of one molecule through a topology handle (include/mkamd_voxel.h (3c)) against the plain
    call on the sigma matrix repeated per frame -- cfg4's shape at a reduced count (periodic, 30 000 atoms, 48^3), open frames, a
    molecule with wide sigmas (the exact fix-up reads the handle's sigmas) and two channel groups: every bit the same; pipelined by
    promise like the plain call; an item of the wrong length is detected on the device; sigmas that the caller changes after the
    handle was built do not matter (the library keeps its own copy); what a topology call cannot serve is refused."""
    import torch
    from moleculekit_amd import _lib, batch
    from tests.synth import synth_config, synth_sigmas
    dev = torch.device("cuda", hip_ctx.device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    cases = []
    p = synth_config(4, 12)                                            # twelve cfg4 frames
    n = int(p["atom_offsets"][1])
    cases.append(("cfg4", p["coords"], p["sigmas"][:n], 12, grid_origin(p["centers"][0], p["boxsize"], 1.0), p["box"], np.float64))
    rng = np.random.default_rng(15)
    n2, F2 = 4000, 7
    sig = synth_sigmas(rng, n2)
    sig[::9, 6] = 2.27
    sig = np.concatenate([sig, sig[:, :3] * 0.8], axis=1)              # 11 channels, wide sigmas, atoms with several sigmas
    base = rng.uniform(-14, 14, size=(n2, 3))
    xyz = np.concatenate([base + rng.normal(0, 0.5, size=(n2, 3)) for _ in range(F2)]).astype(np.float32)
    cases.append(("wide11", xyz, sig, F2, (np.array([-13.0, -12.0, -13.5]), np.array([26, 25, 27])), None, np.float32))
    for name, coords, sig1, F, (o, nv), box, sdt in cases:
        n = sig1.shape[0]
        d_xyz, d_offs = t(coords, np.float32), t(np.arange(F + 1) * n, np.int64)
        d_org = t(np.tile(o, (F, 1)), np.float64)
        d_box = None if box is None else t(box, np.float32)
        d_sig1 = t(sig1, sdt)
        plain = batch.voxelize_lattice_torch(d_xyz, d_offs, d_sig1.repeat(F, 1).contiguous(), d_org, nv, 1.0, box=d_box, ctx=hip_ctx)
        topo = _lib.Topology(hip_ctx, d_sig1, 1.0)
        assert topo.has_wide_sigmas == (name == "wide11") and topo.n_atoms == n
        d_sig1.zero_()                                                 # the caller's sigmas may change or go away: the handle has its own
        got = batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, 1.0, box=d_box, ctx=hip_ctx, topology=topo)
        assert torch.equal(plain, got), name
        # promised (pipelined) calls, back to back, and from a host-side handle
        topo_h = _lib.Topology(hip_ctx, np.asarray(sig1, dtype=sdt), 1.0)
        before = hip_ctx.pipelined_calls()
        outs = []
        for _ in range(3):
            hip_ctx.promise_inputs(None)
            outs.append(batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, 1.0, box=d_box, ctx=hip_ctx, topology=topo_h))
        assert all(torch.equal(plain, x) for x in outs), name
        if coords.shape[0] >= 200000:
            assert hip_ctx.pipelined_calls() > before
        # the wrong split of the right total: flagged by the device, reported at the next synchronize
        bad = np.arange(F + 1) * n
        bad[1] -= 5
        batch.voxelize_lattice_torch(d_xyz, t(bad, np.int64), None, d_org, nv, 1.0, box=d_box, ctx=hip_ctx, topology=topo)
        with pytest.raises(ValueError, match="topology"):
            hip_ctx.synchronize()
        with pytest.raises(ValueError, match="atom count"):             # ... and the wrong total on the host already
            batch.voxelize_lattice_torch(d_xyz[:-3], d_offs, None, d_org, nv, 1.0, box=d_box, ctx=hip_ctx, topology=topo)
        with pytest.raises(ValueError, match="voxel size"):
            batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, 0.5, box=d_box, ctx=hip_ctx, topology=topo)
        hip_ctx.set_force_general(True)
        try:
            with pytest.raises(ValueError, match="class-sorted path only"):
                batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, 1.0, box=d_box, ctx=hip_ctx, topology=topo)
        finally:
            hip_ctx.set_force_general(False)
        assert torch.equal(batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, 1.0, box=d_box, ctx=hip_ctx, topology=topo), plain)
        topo.close(); topo_h.close()
    with pytest.raises(ValueError, match="15 distinct"):               # no class ids to reuse
        _lib.Topology(hip_ctx, np.tile(np.linspace(1.0, 2.9, 20)[:, None], (1, 8)), 1.0)


def test_streamed_and_sharded_frame_drivers_use_the_topology_and_stay_bitwise(hip_ctx):
    """iterVoxelizeTrajectory and ShardedVoxelizer(shared_sigmas=True) build a topology handle for their molecule by construction:
    the same features as with the handle switched off (the sigma matrix repeated per frame), bit for bit."""
    import torch
    from moleculekit_amd import batch
    from moleculekit_amd.distributed import ShardedVoxelizer
    from tests.synth import synth_config
    p = synth_config(4, 10)
    n = int(p["atom_offsets"][1])
    coords = np.ascontiguousarray(np.transpose(p["coords"].reshape(10, n, 3), (1, 2, 0)))          # Molecule.coords layout [N, 3, F]
    box = np.ascontiguousarray(p["box"].T)
    kw = dict(box=box, chunk=4, ctx=hip_ctx)
    center, boxsize = p["centers"][0], p["boxsize"]
    with_topo = torch.cat([f for _, f in batch.iterVoxelizeTrajectory(coords, p["sigmas"][:n], center, boxsize, 1.0, **kw)])
    batch.USE_TOPOLOGY = False
    try:
        without = torch.cat([f for _, f in batch.iterVoxelizeTrajectory(coords, p["sigmas"][:n], center, boxsize, 1.0, **kw)])
    finally:
        batch.USE_TOPOLOGY = True
    assert torch.equal(with_topo, without) and float(with_topo.max()) > 0.5
    o, nv = grid_origin(center, boxsize, 1.0)
    origins = np.tile(o, (10, 1))
    dev = torch.device("cuda", hip_ctx.device)
    sv = ShardedVoxelizer.from_host(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0, box=p["box"], device=dev, ctx=hip_ctx,
                                    shared_sigmas=True)
    assert sv._topo is not None and tuple(sv._d["sigmas"].shape) == (n, 8)
    plain = ShardedVoxelizer.from_host(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, 1.0, box=p["box"], device=dev, ctx=hip_ctx)
    assert plain._topo is None
    a, b = sv.voxelize(), plain.voxelize()
    assert torch.equal(a, b)
    # (ten frames in one call take the depth-8 tiles, the streamed chunks of four the depth-4 ones: same values to float32 noise)
    assert float((a - with_topo).abs().max()) <= 1e-5
    with pytest.raises(ValueError, match="sigma rows differ"):
        sg = p["sigmas"].copy(); sg[n + 3, 7] = 1.23
        ShardedVoxelizer.from_host(p["coords"], p["atom_offsets"], sg, origins, nv, 1.0, box=p["box"], device=dev, ctx=hip_ctx, shared_sigmas=True)


def test_clock_probe_reports_a_plausible_shader_clock():
    """mkamd_clock_probe_dev: one wave counts shader clock ticks against the 100 MHz reference counter on a stream of the caller's
    (bench.py reports the clock the device sustains under the headline load).  Idle device, 20 ms: the reference ticks are what was
    asked for, the clock is a number an MI355X can have."""
    import torch
    from moleculekit_amd import _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    st = torch.cuda.Stream(dev)
    ticks = torch.zeros(2, dtype=torch.int64, device=dev)
    ctx.clock_probe_dev(st.cuda_stream, 20000, ticks.data_ptr())
    st.synchronize()
    t_sh, t_ref = (int(v) for v in ticks.cpu().tolist())
    assert 2_000_000 <= t_ref < 2_200_000, t_ref                      # 20 ms of a 100 MHz counter
    ghz = t_sh / t_ref * 0.1
    assert 0.3 < ghz < 3.0, ghz
    with pytest.raises(Exception):
        ctx.clock_probe_dev(st.cuda_stream, 0, ticks.data_ptr())



def test_random_topology_calls_with_wide_atoms_split_fixup_equals_k_tail_equals_plain():
    """Round 6 (late): a topology call whose molecule has wide sigmas re-decides its cut-off shell in launches of its own (k_exact_shells lists
    the hits, k_exact_redo recomputes each with a wave per 2 048-atom slice, joined by an atomic maximum).  Sixty random calls of
    tests/sweep_gpu_topology.py (50 ... 7 000 atoms, wide atoms on lattice points nudged by ulps: hundreds of hits per call): the split form,
    the hits recomputed inside k_tail and the plain call agree bit for bit (the evidence session runs 1 000; profiles/r6_random_sweep_topology.txt: 6 000)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "sweep_gpu_topology.py"), "100000", "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "60 of 60 calls bit-identical" in r.stdout
