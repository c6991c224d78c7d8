#!/usr/bin/env python3
"""Golden fixtures for the XTC decoder, from the REAL reference reader
(moleculekit/fileformats/xtc: `read_xtc` / `read_xtc_frames`).  Run in the build container only:

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_xtc.py

Stores DATA: trajectory files the reference's own tests hold (whole, or their first frames -- XTC frames are
self-contained records, so a prefix of the file is a valid file) under tests/golden/xtc/, and what the reference
decodes from exactly those files (coords float32 [N,3,F] in nm, box vectors [3,3,F], time, step)."""
import os
import struct
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xtc")
sys.path.insert(0, REF_BUILD)
from moleculekit.xtc import read_xtc, read_xtc_frames  # noqa: E402

SOURCES = {   # name -> (reference test file, frames to keep; None = all)
    "mol": ("/root/reference/tests/test_writers/mol.xtc", None),
    "aladipep": ("/root/reference/tests/test_readers/aladipep_traj_4fs_100ps.xtc", None),
    "3ptb_traj_head": ("/root/reference/tests/test_molecule/3ptb_traj.xtc", 6),
    "4rws_head": ("/root/reference/tests/test_readers/4RWS/traj.xtc", 1),
}


def frame_ends(buf):
    """Byte offsets of the frame ends (XTC record layout; used only to cut a prefix)."""
    ends, p = [], 0
    while p + 16 <= len(buf):
        magic, natoms = struct.unpack(">ii", buf[p:p + 8])
        assert magic == 1995
        p += 16 + 36 + 4
        if natoms <= 9:
            p += 12 * natoms
        else:
            p += 4 + 12 + 12 + 4
            nbytes, = struct.unpack(">i", buf[p:p + 4])
            p += 4 + ((nbytes + 3) // 4) * 4
        ends.append(p)
    return ends


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, (src, keep) in SOURCES.items():
        buf = open(src, "rb").read()
        if keep is not None:
            buf = buf[:frame_ends(buf)[keep - 1]]
        fn = os.path.join(OUT, name + ".xtc")
        open(fn, "wb").write(buf)
        coords, box, time, step = read_xtc(fn.encode("UTF-8"))
        sel = np.array([coords.shape[2] - 1, 0], dtype=np.int32) if coords.shape[2] > 1 else np.array([0], dtype=np.int32)
        c2, b2, t2, s2 = read_xtc_frames(fn.encode("UTF-8"), sel)
        stride = 8 if coords.shape[0] > 20000 else 1          # big system: every 8th atom + a checksum over all bits
        np.savez_compressed(os.path.join(OUT, name + "_decoded.npz"), coords=coords[::stride], stride=np.int64(stride),
                            natoms=np.int64(coords.shape[0]), bitsum=np.uint64(coords.view(np.uint32).astype(np.uint64).sum()),
                            box=box, time=time, step=step, sel=sel, sel_coords=c2[::stride], sel_box=b2, sel_time=t2, sel_step=s2)
        for junk in os.listdir(OUT):                          # the reference reader drops frame-offset caches next to the file
            if junk.startswith(".") and junk.endswith(".xtc"):
                os.remove(os.path.join(OUT, junk))
        print(name, len(buf), "bytes ->", coords.shape, "frames sel", sel)


if __name__ == "__main__":
    main()
