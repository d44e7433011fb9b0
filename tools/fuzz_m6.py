"""Randomised parity sweep of the background matchers (Fuse x2, SearchBySim3, SearchForTriangulation, SearchByBoW(KF,KF),
SearchForInitialization): the scenario builders of tests/test_gpu_matcher_m6.py with random parameters.
    python tools/fuzz_m6.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam2_map_amd import lib as G
from oracle import oracle_py as O
import test_gpu_matcher_m6 as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
count = {}
while time.time() - t0 < budget:
    k = int(rng.integers(0, 6))
    try:
        if k == 0:
            args = (float(rng.choice([1.5, 2.5, 3.0, 4.0, 6.0])), int(rng.integers(0, 1 << 20)))
            T.test_fuse_candidates(G, O, *args)
        elif k == 1:
            args = (float(rng.choice([3.0, 4.0, 10.0])), float(rng.choice([0.6, 0.8, 1.0, 1.3, 1.7])))
            T.test_fuse_sim3_candidates(G, O, *args)
        elif k == 2:
            args = (float(rng.choice([5.0, 7.5, 15.0])), float(rng.choice([0.9, 1.0, 1.15])), float(rng.choice([0.0, 0.1, 0.3])))
            T.test_search_by_sim3(G, O, *args)
        elif k == 3:
            args = (bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(1, 5)))
            T.test_search_for_triangulation(G, O, *args)
        elif k == 4:
            args = (float(rng.choice([0.6, 0.75, 0.9])), bool(rng.integers(0, 2)), int(rng.integers(1, 5)))
            T.test_search_by_bow_keyframes(G, O, *args)
        else:
            args = (int(rng.choice([20, 40, 100, 150])), float(rng.choice([0.7, 0.8, 0.9])), bool(rng.integers(0, 2)),
                    bool(rng.integers(0, 2)))
            T.test_search_for_initialization(G, O, *args)
    except AssertionError as ex:
        print("FAIL", k, args, str(ex)[:300])
        sys.exit(1)
    count[k] = count.get(k, 0) + 1
print("fuzz ok:", count, "in %.0f s" % (time.time() - t0))
