#!/usr/bin/env python3
"""Writes moleculekit_amd/_residue_tables.py: the residue-name / element tables the reference's
``atomtypingValidityChecks`` consults through ``atomselect("protein")`` and ``autoSegment``
(moleculekit/share/atomselect/atomselect.json, moleculekit/residues.py, moleculekit/periodictable.py,
moleculekit/tools/nonstandard_residues.py).  DATA only; run in the build container:

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_residue_tables.py
"""
import json
import os
import sys

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
sys.path.insert(0, REF_BUILD)
import logging  # noqa: E402

logging.disable(logging.CRITICAL)
from moleculekit.periodictable import METAL_ELEMENTS  # noqa: E402
from moleculekit.residues import CAP_RESIDUE_NAMES, LIPID_RESIDUE_NAMES, WATER_RESIDUE_NAMES  # noqa: E402
from moleculekit.tools.nonstandard_residues import _CANONICAL_RESNAMES  # noqa: E402

sel = json.load(open("/root/reference/moleculekit/share/atomselect/atomselect.json"))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "moleculekit_amd", "_residue_tables.py")
with open(out, "w") as f:
    f.write('"""Residue-name / element tables (data generated from the reference by tests/golden/make_residue_tables.py)."""\n')
    for name, vals in (("PROTEIN_BACKBONE_NAMES", sel["protein_backbone_names"]),
                       ("PROTEIN_TERMINAL_NAMES", sel["protein_terminal_names"]),
                       ("ION_RESNAMES", sel["ion_resnames"]), ("WATER_RESNAMES", sorted(WATER_RESIDUE_NAMES)),
                       ("CAP_RESNAMES", sorted(CAP_RESIDUE_NAMES)), ("LIPID_RESNAMES", sorted(LIPID_RESIDUE_NAMES)),
                       ("METAL_ELEMENTS", sorted(METAL_ELEMENTS)), ("CANONICAL_RESNAMES", sorted(_CANONICAL_RESNAMES))):
        f.write(f"{name} = frozenset({list(vals)!r})\n")
print(open(out).read()[:300])
