"""Per-atom property channels from AutoDock (PDBQT) atom types -- the table-driven part of the reference's
``getChannels``, so that ``getVoxelDescriptors(mol)`` works without ``userchannels``
for molecules that already carry atom types (e.g. read from a .pdbqt file).

Mirrors, array in / array out and vectorised (the reference loops over hydrogens and carbons in Python):

* ``atomtype_properties_pdbqt``  = ``_getAtomtypePropertiesPDBQT`` + ``_findDonors``
  (``moleculekit/tools/voxeldescriptors.py:409-509``; ``getChannels(version=1)``)
* ``features_from_atomtypes``    = ``getFeatures`` and its helpers (``moleculekit/tools/atomtyper.py:376-554``;
  the table stage of ``getChannels(version=2)``)
* ``getChannels``                = ``moleculekit/tools/voxeldescriptors.py:135-194`` for ``Molecule``-like objects

The rules around the typing proper (OpenBabel type -> PDBQT type, the validity checks) are in
``moleculekit_amd.atomtyper``.  Not here: OpenBabel's own typing / Gasteiger charges and the ``SmallMol`` branch
(RDKit); neither toolkit is part of this package's environment.
``moleculekit_amd.voxeldescriptors.getChannels`` uses this module by default and delegates to an installed
moleculekit only when asked to (``backend="moleculekit"``).

Host-side numpy only: typing is a few table look-ups per atom, it is not on the GPU path.
"""
from __future__ import annotations

import numpy as np

from ._vdw_radii import VDW_RADIUS

# tools/voxeldescriptors.py:18-27
CHANNEL_ORDER = ("hydrophobic", "aromatic", "hbond_acceptor", "hbond_donor", "positive_ionizable",
                 "negative_ionizable", "metal", "occupancies")

# tools/atomtyper.py:16-40
_METAL_ATYPES = ("MG", "ZN", "MN", "CA", "FE", "HG", "CD", "NI", "CO", "CU", "K", "LI",
                 "Mg", "Zn", "Mn", "Ca", "Fe", "Hg", "Cd", "Ni", "Co", "Cu", "Li")
_HIS_NAMES = ("HIS", "HID", "HIE", "HIP", "HSE", "HSD", "HSP")


def _as_str(a):
    return np.asarray(a).astype(str)


def _bonds(bonds, n):
    b = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    if b.size and (b.min() < 0 or b.max() >= n):
        raise ValueError("bond indices out of range")
    return b


def _bonded_to(mask_h, bonds, n):
    """Boolean [n]: atoms that share a bond with an atom selected by ``mask_h``."""
    out = np.zeros(n, dtype=bool)
    if bonds.size:
        out[bonds[mask_h[bonds[:, 0]], 1]] = True
        out[bonds[mask_h[bonds[:, 1]], 0]] = True
    return out


def _first_char(a):
    return np.array([s[:1] for s in a], dtype="U1")


def atomtype_properties_pdbqt(atomtype, element, name, charge, bonds) -> np.ndarray:
    """``getChannels(version=1)``'s boolean channels, columns in ``CHANNEL_ORDER``.

    Atom types are compared upper-cased (``:457``). Donors are the N/O-named partners of atoms whose
    ELEMENT field is ``HD``/``HS`` (``:503-508`` -- the reference keys on ``mol.element`` here, not on the
    atom type, and so do we). Ionizable = sign of the partial charge (``:469-470``)."""
    t = np.char.upper(_as_str(atomtype))
    el, nm = _as_str(element), _as_str(name)
    q = np.asarray(charge, dtype=np.float64)
    n = t.shape[0]
    b = _bonds(bonds, n)
    out = np.zeros((n, len(CHANNEL_ORDER)), dtype=bool)
    out[:, 0] = (t == "C") | (t == "A")
    out[:, 1] = t == "A"
    out[:, 2] = np.isin(t, ("NA", "NS", "OA", "OS", "SA"))
    out[:, 3] = _bonded_to(np.isin(el, ("HD", "HS")), b, n) & np.isin(_first_char(nm), ("N", "O"))
    out[:, 4] = q > 0
    out[:, 5] = q < 0
    out[:, 6] = np.isin(t, ("MG", "ZN", "MN", "CA", "FE"))
    out[:, 7] = ~np.isin(t, ("H", "HS", "HD"))
    return out


def features_from_atomtypes(atomtype, resname, name, bonds) -> np.ndarray:
    """``getFeatures`` (``tools/atomtyper.py:523-554``): boolean [N, 8], columns in ``CHANNEL_ORDER``.

    Exact-case atom types. Ionizable groups are residue/atom-name rules: the side-chain nitrogens of
    ARG/AR0, LYS/LYN and the histidines, the guanidinium carbon of ARG and the aromatic carbons of the
    histidines (positive); the side-chain oxygens and the carboxyl carbon of ASP/ASH and GLU/GLH
    (negative). "Carboxyl / guanidinium carbon" = a ``C``-typed, non-backbone carbon with exactly three
    bond entries (``:413-416``)."""
    t, rn, nm = _as_str(atomtype), _as_str(resname), _as_str(name)
    n = t.shape[0]
    b = _bonds(bonds, n)
    degree = np.bincount(b.ravel(), minlength=n) if b.size else np.zeros(n, dtype=np.int64)
    three = degree == 3
    out = np.zeros((n, len(CHANNEL_ORDER)), dtype=bool)
    out[:, 0] = t == "C"
    out[:, 1] = np.isin(t, ("A", "Na", "Nn"))
    out[:, 2] = np.isin(t, ("OA", "NA", "SA", "Na"))
    out[:, 3] = _bonded_to(np.isin(t, ("HD", "HS")), b, n) & np.isin(_first_char(t), ("N", "O", "S"))
    his = np.isin(rn, _HIS_NAMES)
    side_n = nm != "N"
    out[:, 4] = ((np.isin(rn, ("ARG", "AR0")) & (t == "N") & side_n)
                 | ((rn == "ARG") & (t == "C") & (nm != "C") & three)
                 | (np.isin(rn, ("LYS", "LYN")) & (t == "N") & side_n)
                 | (his & np.isin(t, ("N", "NA", "Nn", "Na")) & side_n)
                 | (his & (t == "A")))
    acid = np.isin(rn, ("ASP", "ASH", "GLU", "GLH"))
    out[:, 5] = acid & (((t == "OA") & (nm != "O")) | ((t == "C") & (nm != "C") & three))
    out[:, 6] = np.isin(t, _METAL_ATYPES)
    out[:, 7] = _first_char(t) != "H"
    return out


def _mol_bonds(mol):
    """Bonds the way the reference's callers see them: ``mol._getBonds()`` (file + guessed bonds) when the
    object has it, else the ``bonds`` array."""
    if hasattr(mol, "_getBonds"):
        return mol._getBonds()
    return getattr(mol, "bonds")


def getChannels(mol, aromaticNitrogen: bool = False, version: int = 2, validitychecks: bool = True):
    """``moleculekit.tools.voxeldescriptors.getChannels`` (``:135-194``) for ``Molecule``-like objects that
    already carry AutoDock atom types: returns ``(channels float64 [N, 8], mol)`` with boolean masks scaled
    by the element's van-der-Waals radius (``:190-193``).

    ``version=1`` uses the PDBQT types/charges as they are; ``version=2`` in the reference first re-types the
    molecule with OpenBabel (``getPDBQTAtomTypesAndCharges``, ``tools/atomtyper.py:330-373``) -- a third-party step
    that is not available here, so the atom types on ``mol`` are taken as given (``aromaticNitrogen`` only affects
    that step).  ``validitychecks`` runs ``atomtyper.atomtypingValidityChecks`` like the reference does before
    typing, when ``mol`` carries the fields the checks read (resid, chain, segid, coords)."""
    for field in ("atomtype", "element", "name"):
        if not hasattr(mol, field):
            raise TypeError(f"getChannels needs a Molecule-like object with a `{field}` array "
                            "(SmallMol / RDKit typing is not part of moleculekit_amd: pass userchannels)")
    if hasattr(mol, "copy"):
        mol = mol.copy()
    atomtype = _as_str(mol.atomtype)
    if not np.any(atomtype != ""):
        raise RuntimeError("the molecule has no atom types; assign PDBQT atom types first or pass userchannels")
    if version == 1:
        mask = atomtype_properties_pdbqt(atomtype, mol.element, mol.name, mol.charge, _mol_bonds(mol))
    elif version == 2:
        if validitychecks and all(hasattr(mol, f) for f in ("resid", "chain", "segid", "coords", "resname", "bonds")):
            from .atomtyper import atomtypingValidityChecks

            atomtypingValidityChecks(mol)
        mask = features_from_atomtypes(atomtype, mol.resname, mol.name, getattr(mol, "bonds"))
    else:
        raise ValueError("version must be 1 or 2")
    radii = np.array([VDW_RADIUS[e] for e in _as_str(mol.element)])          # _getChannelRadii, :117-121
    return radii[:, np.newaxis] * mask.astype(float), mol
