"""Golden vectors for moleculekit_amd.xtc.write_xtc: what the REAL reference reader (moleculekit/fileformats/xtc: read_xtc)
decodes from files this package's writer produced.  Run in the build container only:

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_xtc_writer.py

Stores, per case, the inputs handed to write_xtc and the reference's decode of the file under
tests/golden/xtc_writer_cases.npz (tests/test_xtc.py writes the files again and compares)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild"))
from moleculekit.xtc import read_xtc  # noqa: E402  (the reference)

from moleculekit_amd import xtc  # noqa: E402


def cases():
    rng = np.random.default_rng(404)
    for name, N, F, L in (("small9", 7, 3, 2.0), ("mixed_radix", 250, 3, 6.69), ("negative_origin", 64, 2, 3.3),
                          ("per_axis_fields", 40, 2, 30000.0)):
        x = (rng.uniform(-0.4, 1.0, size=(N, 3, F)) * L).astype(np.float32)
        box = np.zeros((3, 3, F), np.float32)
        box[0, 0] = L; box[1, 1] = 0.9 * L; box[2, 2] = 1.1 * L; box[1, 0] = 0.1 * L
        yield name, x, box, (np.arange(F) * 0.5).astype(np.float32), (np.arange(F) * 7).astype(np.int32)


if __name__ == "__main__":
    out = {}
    for name, x, box, t, st in cases():
        fn = tempfile.mktemp(suffix=".xtc")
        xtc.write_xtc(fn, x, box, t, st)
        c, b, tt, ss = read_xtc(fn.encode("UTF-8"))
        os.unlink(fn)
        out[name + "_coords"], out[name + "_box"], out[name + "_time"], out[name + "_step"] = x, box, t, st
        out[name + "_ref_coords"], out[name + "_ref_box"] = np.asarray(c), np.asarray(b)
        out[name + "_ref_time"], out[name + "_ref_step"] = np.asarray(tt), np.asarray(ss)
        print(name, x.shape, "max |decoded - written|", float(np.abs(np.asarray(c) - x).max()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "xtc_writer_cases.npz"), **out)
