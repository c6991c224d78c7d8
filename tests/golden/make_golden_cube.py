#!/usr/bin/env python3
"""Golden fixture for the .cube export, from the REAL reference's `writeCube`
(moleculekit/util.py:415-458).  Run in the build container only:

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_cube.py

Stores DATA: a seeded float array (with the value classes a voxel grid holds: exact 0 and 1, tiny tails, ordinary
values), the grid origin / resolution, and the text file the reference writes for it."""
import os
import sys
import tempfile

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)
from moleculekit.util import readCube, writeCube  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    arr = rng.random((5, 4, 7))
    arr[0, 0, :3] = [0.0, 1.0, 1.234e-7]
    arr[1, 2, 3] = 9.87654321e-5
    arr[4, 3, 6] = 0.999999
    vmin, vres = np.array([-9.5, 3.25, 11.0]), np.array([1.0, 0.5, 0.7])
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "ref.cube")
        writeCube(arr, fn, vmin, vres)
        text = open(fn).read()
        back, meta = readCube(fn)
    np.savez_compressed(os.path.join(OUT, "cube_case.npz"), arr=arr, vecMin=vmin, vecRes=vres,
                        text=np.array(text), readback=back, org=np.array(list(meta["org"])))
    print(len(text), "bytes;", text.splitlines()[2])


if __name__ == "__main__":
    main()
