#!/usr/bin/env python3
"""tools/bench_calculate_occupancy.py -- the literal a1 call, calculate_occupancy(centers, coords, sigmas, results) with
the reference's in-place max contract, on the 3PTB grid: lattice centres (what the reference's own caller passes) and
arbitrary ones.  MKAMD_LIB selects the build."""
import os
import numpy as np, time, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from moleculekit_amd.occupancy_utils import calculate_occupancy
from moleculekit_amd.voxeldescriptors import getCenters
g=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'cfg1_3ptb.npz'))
g={k:np.ascontiguousarray(g[k]) for k in ('coords','sigmas','center','features')}      # (an NpzFile re-reads the zip on every access)
c,_=getCenters(boxsize=[24,24,24], center=g['center'], voxelsize=1)
res=np.zeros((c.shape[0],8)); calculate_occupancy(c,g['coords'],g['sigmas'],res)
print('lattice err', np.abs(res-g['features']).max())
j=c+np.random.default_rng(0).normal(0,1e-3,c.shape)
for name,cc in (('lattice',c),('arbitrary',j)):
    r=np.zeros((c.shape[0],8))
    for _ in range(10): calculate_occupancy(cc,g['coords'],g['sigmas'],r)
    t0=time.perf_counter()
    for _ in range(100): calculate_occupancy(cc,g['coords'],g['sigmas'],r)
    print(name,(time.perf_counter()-t0)/100*1e3,'ms per call')
# in-place max semantics
r=np.full((c.shape[0],8),0.5); calculate_occupancy(c,g['coords'],g['sigmas'],r); print('in-place max ok', np.allclose(r, np.maximum(0.5,g['features']), atol=1e-5))
# the 64^3 x 50 000-atom grid of BASELINE.json configs[1] through the same literal call
from tests.synth import synth_config
p = synth_config(2, 1)
c2 = np.ascontiguousarray(p["centers"][0] - p["boxsize"] / 2 + 0.0)            # bb_min of the grid
cen, _ = getCenters(boxsize=list(p["boxsize"]), center=p["centers"][0], voxelsize=float(p["voxelsize"]))
xyz = np.ascontiguousarray(p["coords"], np.float32); sg = np.ascontiguousarray(p["sigmas"], np.float64)
r = np.zeros((cen.shape[0], sg.shape[1]))
for _ in range(2): calculate_occupancy(cen, xyz, sg, r)
t0 = time.perf_counter()
for _ in range(5): calculate_occupancy(cen, xyz, sg, r)
print('cfg2 grid (64^3, 50k atoms), lattice centres:', (time.perf_counter() - t0) / 5 * 1e3, 'ms per call')
