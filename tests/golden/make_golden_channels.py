#!/usr/bin/env python3
"""Golden fixture for the channel-typing row, from the REAL reference.

Run in the build container only (see make_golden.py for the one-off reference build):

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_channels.py

Stores DATA only: the per-atom fields of the reference's own test molecule (3ptb.pdbqt: AutoDock atom
types, charges, elements, names, residue names, bonds as `Molecule._getBonds()` returns them) and what the
reference's table-driven typing makes of them: `_getAtomtypePropertiesPDBQT` (getChannels version=1,
tools/voxeldescriptors.py:409-485), `getFeatures` (the table stage of version=2, tools/atomtyper.py:523-554)
and the sigma channels `getChannels` derives from them (radii x mask, :190-193).
"""
import os
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
REF_TESTS = "/root/reference/tests/test_voxeldescriptors"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)

from moleculekit.molecule import Molecule  # noqa: E402
from moleculekit.tools import atomtyper  # noqa: E402
from moleculekit.tools.voxeldescriptors import (  # noqa: E402
    _getAtomtypePropertiesPDBQT, _getChannelRadii)


def main():
    mol = Molecule(os.path.join(REF_TESTS, "3ptb.pdbqt"))
    bonds = mol._getBonds()
    props_v1 = _getAtomtypePropertiesPDBQT(mol)
    # getChannels itself cannot be imported here (it imports SmallMol -> RDKit, absent from the image): its last
    # step (tools/voxeldescriptors.py:190-193) is applied with the reference's own radii function instead
    radii = _getChannelRadii(mol.get("element"))
    channels_v1 = radii[:, np.newaxis] * np.asarray(props_v1).astype(float)
    mol2 = mol.copy()
    mol2.bonds = bonds.copy()                      # getFeatures reads mol.bonds (tools/atomtyper.py:548)
    feats_v2 = atomtyper.getFeatures(mol2)
    np.savez_compressed(
        os.path.join(OUT, "channels_3ptb.npz"),
        atomtype=mol.atomtype.astype("U4"), charge=mol.charge.astype(np.float32), element=mol.element.astype("U2"),
        name=mol.name.astype("U4"), resname=mol.resname.astype("U4"), bonds=np.asarray(bonds, dtype=np.int64),
        coords=mol.coords[:, :, 0].astype(np.float32),
        props_v1=np.asarray(props_v1, dtype=bool), channels_v1=np.asarray(channels_v1, dtype=np.float64),
        feats_v2=np.asarray(feats_v2, dtype=bool), radii=np.asarray(radii, dtype=np.float64))
    # ---- a small peptide WITH hydrogens (tests/test_writers/mol.pdbqt; its type column is a placeholder), typed
    #      here by a simple deterministic rule so that the donor rules see HD hydrogens: the INPUT typing is
    #      synthetic, the outputs are the reference's.  v1 looks for donors through mol.ELEMENT == "HD"/"HS"
    #      (tools/voxeldescriptors.py:504), v2 through the atom types (tools/atomtyper.py:388-397).
    pep = Molecule("/root/reference/tests/test_writers/mol.pdbqt")
    pbonds = pep._getBonds()
    el = np.array([n[0] for n in pep.name], dtype=object)
    nbr = [[] for _ in range(pep.numAtoms)]
    for a, b in pbonds:
        nbr[a].append(b); nbr[b].append(a)
    at = np.empty(pep.numAtoms, dtype=object)
    ring = {"CG", "CD1", "CD2", "CE1", "CE2", "CZ"}
    for i in range(pep.numAtoms):
        if el[i] == "H":
            at[i] = "HD" if any(el[j] in ("N", "O") for j in nbr[i]) else "H"
            if any(el[j] == "S" for j in nbr[i]):
                at[i] = "HS"
        elif el[i] == "C":
            at[i] = "A" if (pep.resname[i] == "TYR" and pep.name[i] in ring) else "C"
        elif el[i] == "N":
            at[i] = "NA" if i % 3 == 0 else "N"
        elif el[i] == "O":
            at[i] = "OA"
        elif el[i] == "S":
            at[i] = "SA"
        else:
            at[i] = str(el[i])
    pep.atomtype[:] = at
    pep.element[:] = el
    pep_v2 = pep.copy(); pep_v2.bonds = pbonds.copy()
    pfeats_v2 = atomtyper.getFeatures(pep_v2)
    pep_v1 = pep.copy()
    hd = np.isin(at, ("HD", "HS"))
    pep_v1.element[hd] = at[hd]                  # what _findDonors keys on
    pprops_v1 = _getAtomtypePropertiesPDBQT(pep_v1)
    np.savez_compressed(
        os.path.join(OUT, "channels_peptide.npz"),
        atomtype=pep.atomtype.astype("U4"), charge=pep.charge.astype(np.float32), element_v2=pep.element.astype("U2"),
        element_v1=pep_v1.element.astype("U2"), name=pep.name.astype("U4"), resname=pep.resname.astype("U4"),
        bonds=np.asarray(pbonds, dtype=np.int64), props_v1=np.asarray(pprops_v1, dtype=bool),
        feats_v2=np.asarray(pfeats_v2, dtype=bool))
    print("peptide atoms", pep.numAtoms, "v1", np.asarray(pprops_v1).sum(0), "v2", np.asarray(pfeats_v2).sum(0))
    print("atoms", mol.numAtoms, "bonds", bonds.shape, "v1 true per channel", np.asarray(props_v1).sum(0),
          "v2 true per channel", np.asarray(feats_v2).sum(0))


if __name__ == "__main__":
    main()
