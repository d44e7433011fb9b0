"""Randomised parity sweep of SearchByProjection(Frame, MapPoints) -- candidate lists + the blocked Gauss-Seidel claim
fixpoint -- against the oracle: random local maps (1..8 earlier frames, random radius factor, ratio, share of map points
that do not block, pre-associated key points, shuffled map order).
    python tools/fuzz_projection.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam2_map_amd import lib as G
from oracle import oracle_py as O
import test_gpu_matcher_proj as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    w, h, nfeat = [(640, 480, 1000), (1280, 960, 2000), (640, 480, 500), (752, 480, 1500)][int(rng.integers(0, 4))]
    nprev = int(rng.integers(1, 9))
    seed = int(rng.integers(0, 1 << 30))
    obs0 = float(rng.choice([0.0, 0.1, 0.5]))
    st, Tcw, gf, of, mp, *_ = T.build_c3(G, O, w, h, nfeat, nprev, seed, obs_zero_frac=obs0)
    if rng.integers(0, 2):  # visiting order of the map points = claim order
        perm = rng.permutation(len(mp["in_view"]))
        mp = {k: (np.ascontiguousarray(v[perm]) if isinstance(v, np.ndarray) and len(v) == len(perm) else v) for k, v in mp.items()}
    k0 = np.full(gf.n, -1, np.int32)
    if rng.integers(0, 2):
        pre = rng.choice(gf.n, int(rng.integers(1, gf.n // 2)), replace=False)
        k0[pre] = rng.choice([-2, 0], len(pre))
    for _ in range(2):
        th, ratio = float(rng.choice([1.0, 3.0, 5.0, 8.0])), float(rng.choice([0.6, 0.7, 0.8, 0.9]))
        ng, kg = G.ORBmatcher(ratio).SearchByProjection(gf, mp, th, k0)
        no, ko = O.search_by_projection(of, mp, th, ratio, k0)
        if ng != no or not np.array_equal(kg, ko):
            print("FAIL", dict(w=w, h=h, nfeat=nfeat, nprev=nprev, seed=seed, obs0=obs0, th=th, ratio=ratio), ng, no)
            sys.exit(1)
        n += 1
print("fuzz ok: %d searches in %.0f s" % (n, time.time() - t0))
