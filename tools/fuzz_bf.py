"""Randomised parity sweep of the brute-force / vocabulary-guided matchers against the oracle: descriptor sets drawn
from small pools (many near-duplicates -> long claim chains), random sizes, thresholds, ratios, validity masks and
node labels.
    python tools/fuzz_bf.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam2_map_amd import lib as G
from oracle import oracle_py as O
from test_gpu_matcher_bf import low_entropy

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    na, nb = int(rng.integers(1, 1300)), int(rng.integers(1, 1300))
    pool = int(rng.choice([4, 20, 100, 1000]))
    flip = float(rng.choice([0.0, 0.01, 0.03, 0.1]))
    base = rng.integers(0, 256, (pool, 32), dtype=np.uint8)

    def draw(m):
        d = base[rng.integers(0, pool, m)].copy()
        bits = np.unpackbits(d, axis=1)
        bits ^= (rng.random(bits.shape) < flip).astype(np.uint8)
        return np.packbits(bits, axis=1)
    a, b = draw(na), draw(nb)
    aa, ab = (rng.random(na) * 360).astype(np.float32), (rng.random(nb) * 360).astype(np.float32)
    ratio, th = float(rng.choice([0.6, 0.7, 0.9, 10.0])), int(rng.choice([30, 50, 100, 256]))
    ori = bool(rng.integers(0, 2))
    valid = (rng.random(na) < rng.choice([0.5, 0.9, 1.0])).astype(np.uint8)
    m = G.ORBmatcher(ratio, ori)
    ng, mg = m.MatchBruteForce(a, aa, b, ab, valid_a=valid, th_low=th)
    no, mo = O.match_bf(a, aa, b, ab, valid_a=valid, th_low=th, nnratio=ratio, check_orientation=ori)
    if ng != no or not np.array_equal(mg, mo):
        print("FAIL bf", dict(na=na, nb=nb, pool=pool, flip=flip, ratio=ratio, th=th, ori=ori), ng, no)
        sys.exit(1)
    n += 1
print("fuzz ok: %d matches in %.0f s" % (n, time.time() - t0))
