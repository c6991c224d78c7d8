"""Host-side profile of the drop-in call (tools/, GPU box): where the ~0.2 ms of one synchronous
getVoxelDescriptors call goes.  python tools/profile_dropin.py"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg1_3ptb.npz"))
kw = dict(boxsize=[24, 24, 24], center=g["center"], voxelsize=1, usercoords=g["coords"], userchannels=g["sigmas"])
for _ in range(5):
    getVoxelDescriptors(None, **kw)
t0 = time.perf_counter()
for _ in range(200):
    getVoxelDescriptors(None, **kw)
print("per call us", (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    getVoxelDescriptors(None, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
