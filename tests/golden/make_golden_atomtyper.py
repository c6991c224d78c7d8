#!/usr/bin/env python3
"""Golden vectors for the atom-typing row, from the REAL reference.

Run in the build container only (see make_golden.py for the one-off reference build):

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_atomtyper.py

Writes DATA only (inputs and what the reference returns for them):

* ``atomtyper_1atl.npz`` -- the metallo-protein pair the reference's own test holds
  (tests/test_voxeldescriptors.py:109-131: ``1ATL_atomtyped.psf/.pdb`` -> ``1ATL_channels.npy``): the per-atom
  fields of the typed molecule, the stored channel matrix (metals in channel 6), ``getFeatures`` of it, and
  ``getPDBQTAtomType`` (tools/atomtyper.py:43-128) evaluated by the reference on Sybyl-style input types at
  every atom of that molecule: one consistent typing (the inverse image of the stored PDBQT types) plus
  seeded random draws from OpenBabel's type vocabulary, for both values of ``aromaticNitrogen``.  OpenBabel
  itself is not in the image: the INPUT types are synthetic, the outputs are the reference's.
* ``atomtyper_validity.json`` + ``atomtyper_validity_mols.npz`` -- ``atomtypingValidityChecks``
  (tools/atomtyper.py:244-327) on the prepared 1ATL molecule and on nine damaged variants of a 4-chain
  fragment of it: which exception type and message the reference raises (or none).
"""
import json
import os
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
REF_TESTS = "/root/reference/tests/test_voxeldescriptors"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)

import logging  # noqa: E402

logging.disable(logging.CRITICAL)

from moleculekit.molecule import Molecule  # noqa: E402
from moleculekit.tools import atomtyper  # noqa: E402
from moleculekit.tools.voxeldescriptors import _getChannelRadii  # noqa: E402

# OpenBabel's Sybyl-style internal types by element (the vocabulary getPDBQTAtomType is written against)
VOCAB = {
    "C": ["C3", "C2", "C1", "Car", "Cac", "C+"],
    "N": ["N3", "N2", "N1", "Nar", "Nam", "Npl", "N3+", "Ng+", "Nox", "Ntr"],
    "O": ["O3", "O2", "O-", "O.co2", "Oco2"],
    "S": ["S3", "S2", "Sox", "Sac", "So2"],
    "H": ["H", "HC", "HO"],
    "Zn": ["Zn", "ZN"],
    "Ca": ["Ca", "CA"],
}
ODD = ["P", "Pac", "Pox", "F", "Cl", "Br", "I", "Si", "Du", "Xx", "Mg", "MG", "Fe", "K", "LI", "Se"]


def fields(mol):
    return dict(name=mol.name.astype("U4"), resname=mol.resname.astype("U4"), element=mol.element.astype("U2"),
                resid=mol.resid.astype(np.int64), insertion=mol.insertion.astype("U1"), chain=mol.chain.astype("U2"),
                segid=mol.segid.astype("U4"), bonds=np.asarray(mol.bonds, dtype=np.int64),
                bondtype=mol.bondtype.astype("U2"), coords=mol.coords[:, :, 0].astype(np.float32))


def inverse_typing(mol):
    """One Sybyl-style typing whose image under getPDBQTAtomType is the molecule's stored PDBQT typing."""
    out = np.empty(mol.numAtoms, dtype=object)
    for i, t in enumerate(mol.atomtype):
        el = mol.element[i]
        if t == "A":
            out[i] = "Car"
        elif t == "C":
            out[i] = "C3"
        elif t == "NA":
            out[i] = "Nam" if len(np.where(mol.bonds == i)[0]) == 2 else "N2"
        elif t == "N":
            out[i] = "N3+" if len(np.where(mol.bonds == i)[0]) == 4 else "Nam"
        elif t == "OA":
            out[i] = "O2"
        elif t == "SA":
            out[i] = "S3"
        elif t in ("H", "HD"):
            out[i] = "H"
        else:
            out[i] = el            # metals
    return out


def run_typing(types, mol, aromaticNitrogen):
    return np.array([atomtyper.getPDBQTAtomType(str(t), i, mol, aromaticNitrogen) for i, t in enumerate(types)]).astype("U4")


def validity(mol):
    try:
        atomtyper.atomtypingValidityChecks(mol)
        return {"raises": None, "message": None}
    except Exception as e:  # noqa: BLE001
        return {"raises": type(e).__name__, "message": str(e)}


def main():
    typed = Molecule(os.path.join(REF_TESTS, "1ATL_atomtyped.psf"))
    typed.read(os.path.join(REF_TESTS, "1ATL_atomtyped.pdb"))
    ref_channels = np.load(os.path.join(REF_TESTS, "1ATL_channels.npy"))
    feats = atomtyper.getFeatures(typed)
    radii = _getChannelRadii(typed.element)
    assert np.array_equal(radii[:, None] * feats.astype(float), ref_channels)      # the fixture pins the table stage

    rng = np.random.default_rng(20260927)
    trials = [inverse_typing(typed)]
    for _ in range(5):
        t = np.empty(typed.numAtoms, dtype=object)
        for i, el in enumerate(typed.element):
            pool = VOCAB[el] if rng.random() < 0.9 else ODD
            t[i] = pool[rng.integers(len(pool))]
        trials.append(t)
    sybyl = np.stack(trials).astype("U6")
    out_plain = np.stack([run_typing(t, typed, False) for t in trials])
    out_arom = np.stack([run_typing(t, typed, True) for t in trials])
    assert np.array_equal(out_plain[0], typed.atomtype.astype("U4"))                # the inverse typing round-trips
    # getPDBQTAtomTypesAndCharges' HIP rule (tools/atomtyper.py:361-366) applied by hand on the inverse typing
    hip = np.array([(typed.resname[i] == "HIP" and typed.name[i].strip().startswith("C")
                     and typed.name[i].strip() not in ("CA", "C", "CB")) for i in range(typed.numAtoms)])
    np.savez_compressed(os.path.join(OUT, "atomtyper_1atl.npz"), atomtype=typed.atomtype.astype("U4"),
                        charge=typed.charge.astype(np.float32), ref_channels=ref_channels, feats_v2=np.asarray(feats, dtype=bool),
                        radii=radii.astype(np.float64), sybyl=sybyl, pdbqt_plain=out_plain, pdbqt_arom=out_arom,
                        hip_ring_carbon=hip, **fields(typed))
    print("1ATL atoms", typed.numAtoms, "channels true per column", (ref_channels != 0).sum(0), "HIP ring carbons", int(hip.sum()))

    # ---- validity checks: the whole prepared molecule, then variants of a fragment (chains 0 + Z: 60 residues + metals)
    prep = Molecule(os.path.join(REF_TESTS, "1ATL_prepared.psf"))
    prep.read(os.path.join(REF_TESTS, "1ATL_prepared.pdb"))
    cases, mols = {}, {}
    cases["prepared_full"] = validity(prep)

    def add(tag, m):
        cases[tag] = validity(m)
        for k, v in fields(m).items():
            mols[f"{tag}__{k}"] = v

    frag = prep.copy()
    keep = ((frag.segid == "P0") & (frag.resid <= np.unique(frag.resid[frag.segid == "P0"])[39])) | (frag.segid == "ME")
    frag.filter(keep, _logger=False)
    add("fragment_ok", frag)
    m = frag.copy(); m.filter(m.element != "H", _logger=False); add("no_hydrogens", m)
    m = frag.copy(); m.bonds = np.vstack([m.bonds, m.bonds[:3][:, ::-1]]); m.bondtype = np.hstack([m.bondtype, m.bondtype[:3]]); add("duplicate_bonds", m)
    m = frag.copy(); m.segid[:] = ""; add("blank_segids", m)
    m = frag.copy(); m.chain[:] = ""; add("blank_chains", m)
    m = frag.copy(); m.bonds = m.bonds[:50]; m.bondtype = m.bondtype[:50]; add("too_few_bonds", m)
    m = frag.copy(); m.resname[m.resid == m.resid[0]] = "HOH"; add("foreign_residue", m)
    m = frag.copy(); m.filter(m.segid == "ME", _logger=False); add("metals_only", m)
    # a residue without a protein backbone (a ligand): its backbone atoms renamed
    m = frag.copy(); r5 = (m.segid == "P0") & (m.resid == np.unique(m.resid[m.segid == "P0"])[5])
    m.resname[r5] = "LIG"
    for old, new in (("N", "NX"), ("CA", "CX"), ("C", "CY"), ("O", "OX")):
        m.name[r5 & (m.name == old)] = new
    add("nonprotein_residue", m)
    # a chain break that the segment ids do not reflect: residues 10..14 of the fragment removed
    rs = np.unique(frag.resid[frag.segid == "P0"])
    m = frag.copy(); m.filter(~((m.segid == "P0") & np.isin(m.resid, rs[10:15])), _logger=False)
    add("unsegmented_gap", m)
    # the same gap with the segments named as autoSegment would name them: passes again
    m2 = m.copy(); late = (m2.segid == "P0") & (m2.resid > rs[14]); m2.segid[late] = "P9"; m2.chain[late] = "9"; add("segmented_gap", m2)
    json.dump(cases, open(os.path.join(OUT, "atomtyper_validity.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "atomtyper_validity_mols.npz"), **mols)
    for k, v in cases.items():
        print(f"{k:18s} {v['raises']}  {(v['message'] or '')[:90]}")


if __name__ == "__main__":
    main()
