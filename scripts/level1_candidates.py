"""CPU count (numpy / scipy, no GPU) of the candidates a level-1 point of the headline's first pass meets: how many scan points fail
level 0 at the perturbed pose, the entries of the eight level-2 voxel lists around each of them, and what is left of those after
pruning with the true 5th-neighbour distance / with the bound a failed level-0 attempt proves (profiles/experiments_r06/octant_level1_ab.txt)."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lvamd; 
from importlib import import_module
import importlib.util, os
spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "limo-velo_amd", "synth.py")); synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
from scipy.spatial import cKDTree
sc = synth.make_scene(1_048_576, 65_536)
M = sc["map_xyz"].astype(np.float32); S = sc["scan_xyz"].astype(np.float32)
x = sc["x_init"]
print(x[:10])
# state layout: pos(3), quat rot (x,y,z,w)?  use the oracle-free approach: R from quaternion at x[3:7]
def q2R(q):
    x_,y_,z_,w_ = q
    return np.array([[1-2*(y_*y_+z_*z_), 2*(x_*y_-z_*w_), 2*(x_*z_+y_*w_)],[2*(x_*y_+z_*w_),1-2*(x_*x_+z_*z_),2*(y_*z_-x_*w_)],[2*(x_*z_-y_*w_),2*(y_*z_+x_*w_),1-2*(x_*x_+y_*y_)]])
R = q2R(x[3:7]); p = x[0:3]
W = (S @ R.T + p).astype(np.float32)
tree = cKDTree(M)
d, _ = tree.query(W, k=5)
d5 = d[:,4]
org = M.min(0) - 1.0
cell=0.5
t = (W - org)/cell
c0 = np.floor(t).astype(np.int64)
# level-0 certified radius
marg = np.minimum(t - c0, 1 - (t - c0)).min(1)
r0 = cell*(1+marg)*0.999
lvl1 = d5 >= r0
print("level-0 fails", lvl1.sum(), "of", len(W))
# level-2 voxel counts
c2m = np.floor((M - org)/(4*cell)).astype(np.int64)
key = (c2m[:,0]<<40)|(c2m[:,1]<<20)|c2m[:,2]
uk, cnt = np.unique(key, return_counts=True)
import collections
dct = dict(zip(uk.tolist(), cnt.tolist()))
Wq = W[lvl1]; tq = t[lvl1]; D = d5[lvl1]**2
c2 = np.floor(tq/4).astype(np.int64)
rel = tq - c2*4
b = c2 - (rel < 2)
tot_all = np.zeros(len(Wq)); tot_pr = np.zeros(len(Wq)); nl = np.zeros(len(Wq))
for i in range(8):
    n = b + np.array([i&1,(i>>1)&1,i>>2])
    k = (n[:,0]<<40)|(n[:,1]<<20)|n[:,2]
    c = np.array([dct.get(int(v),0) for v in k])
    lo = n*4.0
    e = np.maximum(np.maximum(lo - tq, tq-(lo+4)),0)
    d2 = (e**2).sum(1)*cell*cell
    keep = d2 <= D
    tot_all += c; tot_pr += c*keep; nl += (c*keep>0)
print("candidates all 8 lists: mean %.0f p50 %.0f p90 %.0f max %.0f" % (tot_all.mean(), np.median(tot_all), np.percentile(tot_all,90), tot_all.max()))
print("after pruning with the TRUE d5 as bound: mean %.0f p50 %.0f p90 %.0f max %.0f; lists %.2f" % (tot_pr.mean(), np.median(tot_pr), np.percentile(tot_pr,90), tot_pr.max(), nl.mean()))
# block radius
m = np.minimum(tq - b*4, b*4+8 - tq).min(1)*cell*0.999
print("certified by octant block:", (np.sqrt(D) < m).sum(), "of", len(D), " d5 stats", np.percentile(np.sqrt(D),[10,50,90]))
# actual level-0 bound: 5th smallest distance among the points of the 27-block of level-0 voxels around the query's voxel
c0m = np.floor((M - org)/cell).astype(np.int64)
kq = c0[lvl1]
# use KD-tree ball query limited to the block: approximate by querying k=64 nearest and filtering to the block
dd, ii = tree.query(Wq, k=48)
inb = np.all(np.abs(c0m[ii] - kq[:,None,:]) <= 1, axis=2)
bound = np.full(len(Wq), np.inf)
for i in range(len(Wq)):
    v = dd[i][inb[i]]
    if len(v) >= 5: bound[i] = v[4]**2
print("level-0 bound available (within 48 nn):", np.isfinite(bound).mean(), " bound/true d5^2 ratio p50", np.median(bound[np.isfinite(bound)]/D[np.isfinite(bound)]))
tot_b = np.zeros(len(Wq))
for i in range(8):
    n = b + np.array([i&1,(i>>1)&1,i>>2])
    k = (n[:,0]<<40)|(n[:,1]<<20)|n[:,2]
    c = np.array([dct.get(int(v),0) for v in k])
    lo = n*4.0
    e = np.maximum(np.maximum(lo - tq, tq-(lo+4)),0)
    d2 = (e**2).sum(1)*cell*cell
    tot_b += c*(d2 <= bound)
print("after pruning with the LEVEL-0 bound: mean %.0f p50 %.0f p90 %.0f" % (tot_b.mean(), np.median(tot_b), np.percentile(tot_b,90)))
