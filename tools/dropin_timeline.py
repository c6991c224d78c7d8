#!/usr/bin/env python3
"""tools/dropin_timeline.py -- 60 drop-in getVoxelDescriptors(3PTB) calls (host arrays in, float64 out: inputs and result in
mapped pinned memory); run under `rocprofv3 --kernel-trace` and summarised by tools/single_timeline_report.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_3ptb.npz"))
kw = dict(boxsize=[24, 24, 24], center=g["center"], voxelsize=1, usercoords=g["coords"], userchannels=g["sigmas"])
import time
for _ in range(60):
    getVoxelDescriptors(None, **kw)
    time.sleep(0.0005)
