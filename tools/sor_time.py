import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from orb_slam2_map_amd import lib as G
import test_sor as T
rng = np.random.default_rng(3)
for n in (100000, 400000, 1000000):
    xyz = np.vstack([T.surface(n // 2, rng), T.surface(n - n // 2, rng) + np.float32([0.003, 0.002, 0.0])])
    pts = T.cloud(xyz)
    G.statistical_outlier_removal(pts[:1000], 50, 1.0)
    t0 = time.perf_counter()
    k, md = G.statistical_outlier_removal(pts, 50, 1.0)
    t1 = time.perf_counter()
    print(n, "points:", round((t1 - t0) * 1e3, 1), "ms end to end (upload, device search, host statistics, compaction); kept", len(k))
