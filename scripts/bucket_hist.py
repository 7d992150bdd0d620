"""Sizes of the level-0 neighbourhood buckets the benchmark scan visits (CPU only; test infrastructure: uses the oracle transform):
mean 62 candidates, 25 % above the 64 of a chunk, 64 % of the 8-point tasks with at least one such bucket."""
import sys; import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle'))
import numpy as np
import lvamd; lvamd.load()
from limo_velo_amd import synth
import lvoracle as o
sc = synth.make_scene(1_048_576, 65_536)
m = sc["map_xyz"].astype(np.float64)
cell = 0.5
org = m.min(axis=0) - 1.0
vm = np.floor((m - org) / cell).astype(np.int64)
key = (vm[:,0] << 42) | (vm[:,1] << 21) | vm[:,2]
uk, cnt = np.unique(key, return_counts=True)
d = dict(zip(uk.tolist(), cnt.tolist()))
w = o.transform_scan(sc["x_true"], sc["scan_xyz"]).astype(np.float64)
vq = np.floor((w - org) / cell).astype(np.int64)
tot = np.zeros(len(w), np.int64)
for dx in (-1,0,1):
    for dy in (-1,0,1):
        for dz in (-1,0,1):
            k = ((vq[:,0]+dx) << 42) | ((vq[:,1]+dy) << 21) | (vq[:,2]+dz)
            tot += np.array([d.get(int(x),0) for x in k])
print("mean", tot.mean(), "median", np.median(tot), "p10/p90", np.percentile(tot,[10,90]), "frac > 64:", (tot>64).mean(), "frac > 128:", (tot>128).mean())
# tasks of 8 consecutive points in Morton-ish order: approximate with sorting by voxel key
order = np.lexsort((vq[:,2], vq[:,1], vq[:,0]))
t = tot[order][: len(tot)//8*8].reshape(-1,8)
print("tasks with any bucket > 64:", (t.max(axis=1) > 64).mean(), " > 128:", (t.max(axis=1) > 128).mean(), "mean chunks per task", np.ceil(t.max(axis=1)/64).mean())
