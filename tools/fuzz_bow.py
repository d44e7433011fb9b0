"""Randomised parity sweep of the vocabulary transform and the node-wise SearchByBoW against the oracle: random tree
shapes (k 2..20, L 1..6, irregular branching, stopped words), the four weighting and six scoring types, random
levelsup, near-duplicate features.
    python tools/fuzz_bow.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam2_map_amd import lib as G
from oracle import oracle_py as O
import scenario
from test_gpu_bow import _features, _same

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    k, L = int(rng.integers(2, 21)), int(rng.integers(1, 7))
    while k ** L > 200000:
        L -= 1
    irregular, stop = bool(rng.integers(0, 2)), float(rng.choice([0.0, 0.05, 0.3]))
    weighting, scoring = int(rng.integers(0, 4)), int(rng.integers(0, 6))
    cfg = dict(k=k, L=L, irregular=irregular, stop=stop, weighting=weighting, scoring=scoring)
    v = scenario.synthetic_vocabulary(k, L, int(rng.integers(0, 1 << 20)), irregular, stop)
    gv = G.ORBVocabulary(k, L, v["parent"], v["is_leaf"], v["desc"], v["weight"], weighting, scoring)
    ov = O.Vocabulary(k, L, v["parent"], v["is_leaf"], v["desc"], v["weight"], weighting, scoring)
    nf = int(rng.integers(1, 1500))
    seed = int(rng.integers(0, 1 << 20))
    da, db = _features(v, nf, seed, p=float(rng.choice([0.01, 0.06, 0.2]))), _features(v, nf, seed + 1)
    if nf > 10:
        db[: nf // 2] = np.packbits(np.unpackbits(da[rng.permutation(nf)[: nf // 2]], axis=1) ^
                                    (rng.random((nf // 2, 256)) < 0.02).astype(np.uint8), axis=1)
    levelsup = int(rng.integers(0, L + 3))
    try:
        ga, oa = gv.transform(da, levelsup), ov.transform(da, levelsup)
        gb, ob = gv.transform(db, levelsup), ov.transform(db, levelsup)
        _same(ga, oa), _same(gb, ob)
        aa, ab = rng.uniform(0, 360, nf).astype(np.float32), rng.uniform(0, 360, nf).astype(np.float32)
        valid = (rng.random(nf) < 0.9).astype(np.uint8)
        ratio, ori = float(rng.choice([0.6, 0.75, 0.95])), bool(rng.integers(0, 2))
        ng, mg = G.search_by_bow(da, aa, valid, ga["node_id"], db, ab, gb["node_id"], 50, ratio, ori)
        no, mo = O.search_by_bow(da, aa, valid, oa, db, ab, ob, 50, ratio, ori)
        assert ng == no and np.array_equal(mg, mo), "search_by_bow %d vs %d" % (ng, no)
    except AssertionError as ex:
        print("FAIL", cfg, nf, levelsup, str(ex)[:200])
        sys.exit(1)
    gv.close()
    n += 1
print("fuzz ok: %d vocabularies in %.0f s" % (n, time.time() - t0))
