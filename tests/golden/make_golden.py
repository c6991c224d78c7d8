#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (the reference never travels to the GPU box):

    # one-off: build the reference's extensions in a scratch dir OUTSIDE the repo
    mkdir -p /tmp/mkbuild && cd /tmp/mkbuild && cp -r /root/reference/{moleculekit,setup.py,pyproject.toml} .
    MOLECULEKIT_DISABLE_STABLE_ABI=1 python3 setup.py build_ext --inplace -j8
    # then
    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden.py

Every array written here is DATA: seeded synthetic inputs, data files the reference's own tests
hold (tests/test_voxeldescriptors/*.npy, *.mol2 coordinates), and the outputs of the reference's
own compiled code on them (calculate_occupancy, getCenters, getVoxelDescriptors, rotateCoordinates,
boundingBox, dist_trajectory).  No reference source text is stored.
"""
import json
import os
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
REF_TESTS = "/root/reference/tests/test_voxeldescriptors"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)

from moleculekit.molecule import Molecule  # noqa: E402
from moleculekit.occupancy_utils import calculate_occupancy  # noqa: E402
from moleculekit.periodictable import periodictable  # noqa: E402
from moleculekit.tools.voxeldescriptors import (  # noqa: E402
    _getChannelRadii, getCenters, getVoxelDescriptors, rotateCoordinates)
from moleculekit.util import boundingBox  # noqa: E402
from moleculekit import distance_utils  # noqa: E402

sys.path.insert(0, os.path.join(OUT, "..", ".."))
from tests.synth import synth_sigmas, synth_config  # noqa: E402  (shared seeded generators)


def ref_occupancy(centers, coords, sigmas):
    """voxeldescriptors.py:515-533 marshalling + the reference kernel."""
    res = np.zeros((centers.shape[0], sigmas.shape[1]))
    calculate_occupancy(np.ascontiguousarray(centers, np.float64),
                        np.ascontiguousarray(coords, np.float32),
                        np.ascontiguousarray(sigmas, np.float64), res)
    return res


def ref_occupancy_pbc(centers, coords, sigmas, box, kmax=2):
    """shifted calls of the reference kernel max-accumulating into the SAME
    results buffer (legal: occupancy_utils.pyx:61 max-accumulates in place).  27 images
    (kmax=1) suffice for wrapped atoms and a grid inside the box; the fixture below has
    unwrapped atoms and a grid that pokes out of the box, so (2*kmax+1)^3 = 125 images are used."""
    res = np.zeros((centers.shape[0], sigmas.shape[1]))
    coords = np.ascontiguousarray(coords, np.float32)
    sigmas = np.ascontiguousarray(sigmas, np.float64)
    box = np.asarray(box, np.float64)
    ks = range(-kmax, kmax + 1)
    for kx in ks:
        for ky in ks:
            for kz in ks:
                shifted = np.ascontiguousarray(centers - np.array([kx, ky, kz]) * box)
                calculate_occupancy(shifted, coords, sigmas, res)
    return res


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    # ---- vdW radii table (periodictable.py:25-306) --------------------------------------
    radii = {k: v.vdw_radius for k, v in periodictable.items()}
    with open(os.path.join(OUT, "vdw_radii.json"), "w") as f:
        json.dump(radii, f, indent=0, sort_keys=True)

    # ---- cfg1: 3PTB, 24^3 @ 1 A (reference test inputs; BASELINE.json configs[0]) --------
    coords = np.load(os.path.join(REF_TESTS, "3PTB_coords_inp.npy"))
    sigmas = np.load(os.path.join(REF_TESTS, "3PTB_channels_inp.npy"))
    center = coords.mean(0).astype(np.float64)
    feats, centers, nvox = getVoxelDescriptors(
        None, boxsize=[24, 24, 24], center=center, voxelsize=1,
        usercoords=coords, userchannels=sigmas)
    save("cfg1_3ptb.npz", coords=coords, sigmas=sigmas, center=center,
         boxsize=np.array([24, 24, 24]), voxelsize=np.float64(1.0),
         features=feats, centers=centers, nvoxels=nvox)

    # ---- 3ptb.pdbqt through a real Molecule: bbox branch, buffer=8 (test_voxeldescriptors.py:77-79)
    mol = Molecule(os.path.join(REF_TESTS, "3ptb.pdbqt"))
    mol.element[mol.element == "CA"] = "Ca"
    occ = (mol.element != "H")
    rad = _getChannelRadii(mol.element)
    userch = np.zeros((mol.numAtoms, 8), dtype=bool)
    userch[:, 7] = occ
    userch[:, 0] = mol.element == "C"
    userch[:, 2] = (mol.element == "O") | (mol.element == "N")
    feats, centers, nvox = getVoxelDescriptors(mol, buffer=8, voxelsize=1, userchannels=userch)
    rng = np.random.default_rng(77)
    samp = np.sort(rng.choice(centers.shape[0], 20000, replace=False))
    save("3ptb_bbox_buffer8.npz", coords=mol.coords[:, :, 0], element=mol.element.astype("U2"),
         userchannels=userch, radii=rad, buffer=np.float64(8), voxelsize=np.float64(1),
         nvoxels=nvox, centers_first=centers[:4], centers_last=centers[-4:],
         bbox=boundingBox(mol), sample_idx=samp, sample_features=feats[samp],
         channel_sums=feats.sum(0), nonzero=np.count_nonzero(feats, axis=0))

    # ---- reference-held fixtures: celecoxib / ledipasvir (test_voxeldescriptors.py:41-68) --
    for name in ("celecoxib", "ledipasvir"):
        m = Molecule(os.path.join(REF_TESTS, f"{name}.mol2"))
        reff, refc, refn = np.load(os.path.join(REF_TESTS, f"{name}_voxres.npy"), allow_pickle=True)
        c, n = getCenters(m, buffer=1)
        assert np.array_equal(c, refc) and np.array_equal(n, refn)
        ch = np.zeros((m.numAtoms, 8), dtype=bool)
        ch[:, 7] = m.element != "H"
        f7, _, _ = getVoxelDescriptors(m, buffer=1, userchannels=ch)
        assert np.allclose(f7[:, 7], np.asarray(reff, np.float64)[:, 7])
        save(f"{name}_ch7.npz", coords=m.coords[:, :, 0], element=m.element.astype("U2"),
             buffer=np.float64(1), ref_centers=np.asarray(refc, np.float64),
             ref_nvoxels=np.asarray(refn, np.int64),
             ref_features_ch7=np.asarray(reff, np.float64)[:, 7],
             radii=_getChannelRadii(m.element))

    # ---- getCenters cases (voxeldescriptors.py:197-248) ------------------------------------
    gc = {}
    rng = np.random.default_rng(5)
    class _M:  # minimal duck-typed molecule for boundingBox (util.py:376-379)
        def __init__(self, c): self.coords = c[:, :, None]; self.frame = 0
        def get(self, field, sel=None):
            assert field == "coords"; return self.coords[:, :, 0]
    cases = []
    for i, (n, buf, vs) in enumerate([(17, 0, 1), (40, 1, 1), (33, 2.5, 0.5), (12, 0.3, 0.7),
                                      (5, 8, 2), (64, 1.0, 0.25)]):
        c = (rng.normal(size=(n, 3)) * 4 + rng.uniform(-30, 30, 3)).astype(np.float32)
        cen, nv = getCenters(_M(c), buffer=buf, voxelsize=vs)
        gc[f"bbox{i}_coords"] = c
        gc[f"bbox{i}_params"] = np.array([buf, vs], np.float64)
        gc[f"bbox{i}_centers"] = cen
        gc[f"bbox{i}_nvoxels"] = nv
        cases.append(f"bbox{i}")
    for i, (bs, cen0, vs) in enumerate([([24, 24, 24], [0, 0, 0], 1), ([12, 12, 12], [1.5, -2.25, 3.0], 0.5),
                                        ([10, 7, 5], [0.1, 0.2, 0.3], 0.7), ([8.5, 9.5, 3.2], [100.0, -50.0, 7.0], 1.0),
                                        ([20, 20, 20], [39.685, 39.685, 39.685], 2)]):
        cen, nv = getCenters(None, boxsize=bs, center=cen0, voxelsize=vs)
        gc[f"box{i}_boxsize"] = np.array(bs, np.float64)
        gc[f"box{i}_center"] = np.array(cen0, np.float64)
        gc[f"box{i}_voxelsize"] = np.float64(vs)
        gc[f"box{i}_centers"] = cen
        gc[f"box{i}_nvoxels"] = nv
    save("getcenters_cases.npz", **gc)

    # ---- rotateCoordinates (voxeldescriptors.py:78-114) ------------------------------------
    c = rng.normal(size=(25, 3)).astype(np.float32) * 5
    rots = np.array([[0.3, -1.2, 2.0], [0, 0, 0], [np.pi, np.pi / 2, -np.pi / 3]])
    cen = np.array([1.0, -2.0, 0.5])
    save("rotate_cases.npz", coords=c, rotations=rots, center=cen,
         out=np.stack([rotateCoordinates(c, list(r), cen) for r in rots]))

    # ---- cfg3-like / cfg5-like small-molecule batches --------------------------------------
    for cfg, nmol in ((3, 6), (5, 6)):
        p = synth_config(cfg, nmol)
        outs = []
        for b in range(nmol):
            s, e = p["atom_offsets"][b], p["atom_offsets"][b + 1]
            f, cen, nv = getVoxelDescriptors(None, boxsize=list(p["boxsize"]), center=list(p["centers"][b]),
                                             voxelsize=p["voxelsize"], usercoords=p["coords"][s:e],
                                             userchannels=p["sigmas"][s:e])
            outs.append(f)
        save(f"cfg{cfg}_small.npz", nmol=np.int64(nmol), features=np.stack(outs), nvoxels=nv,
             coords=p["coords"], sigmas=p["sigmas"], atom_offsets=p["atom_offsets"],
             centers=p["centers"], boxsize=p["boxsize"], voxelsize=np.float64(p["voxelsize"]))

    # ---- dense mid-size lattice case with non-default sigmas (incl. metals, sigma>1.915) ----
    rng = np.random.default_rng(11)
    N = 4000
    c = rng.uniform(0, 34.2, size=(N, 3)).astype(np.float32)
    s = synth_sigmas(rng, N)
    big = rng.choice(N, 40, replace=False)
    s[big, 6] = rng.choice([2.27, 2.75, 1.73, 2.31], size=40)       # Na, K, Mg, Ca vdW radii
    s[big[:10], 3] = rng.uniform(0.5, 3.0, size=10)                  # arbitrary user floats
    f, cen, nv = getVoxelDescriptors(None, boxsize=[30, 27, 21], center=[17.1, 16.0, 18.3],
                                     voxelsize=1, usercoords=c, userchannels=s)
    save("dense_mixed.npz", coords=c, sigmas=s, boxsize=np.array([30., 27., 21.]),
         center=np.array([17.1, 16.0, 18.3]), voxelsize=np.float64(1), features=f, nvoxels=nv)

    # ---- non-8-channel and explicit-centre cases (calculate_occupancy drop-in) --------------
    rng = np.random.default_rng(12)
    for C in (1, 3, 11):
        N = 300
        c = rng.normal(size=(N, 3)).astype(np.float32) * 6
        s = rng.choice([0, 0, 1.1, 1.7, 1.52, 2.0], size=(N, C)).astype(np.float64)
        centers = rng.uniform(-12, 12, size=(777, 3))
        save(f"explicit_C{C}.npz", coords=c, sigmas=s, centers=centers,
             features=ref_occupancy(centers, c, s))

    # ---- cfg2 (BASELINE.json configs[1]): 50k atoms, 64^3 @ 1 A; sampled voxels + checksums --
    p = synth_config(2, 1)
    f, cen, nv = getVoxelDescriptors(None, boxsize=list(p["boxsize"]), center=list(p["centers"][0]),
                                     voxelsize=p["voxelsize"], usercoords=p["coords"],
                                     userchannels=p["sigmas"])
    rng = np.random.default_rng(202)
    samp = np.sort(rng.choice(f.shape[0], 16384, replace=False))
    save("cfg2_sampled.npz", sample_idx=samp, sample_features=f[samp], channel_sums=f.sum(0),
         nonzero=np.count_nonzero(f, axis=0), channel_max=f.max(0), nvoxels=nv)

    # ---- periodic (cfg4-like, small): 27-image reference composition -----------------------
    rng = np.random.default_rng(4)
    box = np.array([31.3, 29.7, 33.1], np.float32)
    N = 3000
    c = (rng.uniform(0, 1, size=(N, 3)) * box).astype(np.float32)
    c[:200] += (rng.integers(-1, 2, size=(200, 3)) * box).astype(np.float32)  # some unwrapped atoms
    s = synth_sigmas(rng, N)
    cen0 = (box.astype(np.float64) / 2)
    centers, nv = getCenters(None, boxsize=[36, 24, 30], center=list(cen0), voxelsize=1)
    fp = ref_occupancy_pbc(centers, c, s, box)
    save("pbc_small.npz", coords=c, sigmas=s, box=box, boxsize=np.array([36., 24., 30.]),
         center=cen0, voxelsize=np.float64(1), features=fp, nvoxels=nv)

    # ---- min-image primitive (distance_utils.pyx:34-54) via dist_trajectory -----------------
    rng = np.random.default_rng(8)
    F = 4
    xyz = rng.uniform(-40, 40, size=(12, 3, F)).astype(np.float32)
    bx = rng.uniform(11, 30, size=(3, F)).astype(np.float32)
    sel1 = np.arange(0, 6, dtype=np.uint32)
    sel2 = np.arange(6, 12, dtype=np.uint32)
    chains = np.arange(12, dtype=np.uint32)       # all different chains -> pbc applies everywhere
    res = np.zeros((F, 36), dtype=np.float32)
    distance_utils.dist_trajectory(xyz, bx, sel1, sel2, chains, False, True, res)
    save("min_image.npz", coords=xyz, box=bx, sel1=sel1, sel2=sel2, dist=res)


if __name__ == "__main__":
    main()
