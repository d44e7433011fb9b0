"""Randomised parity sweep of the extractor against the oracle (sizes, parameters, image statistics):
    python tools/fuzz_extract.py [seconds] [seed]
Not part of the test suite (it needs a GPU and runs as long as asked); prints the first failing configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream
from oracle import oracle_py as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def image(kind, w, h):
    if kind == 0:
        return Stream(w, h, int(rng.integers(1, 1 << 30))).frame(int(rng.integers(0, 40)))[0]
    if kind == 4:  # textured stream with large flat regions (what the FAST early-out keys on)
        return Stream(w, h, int(rng.integers(1, 1 << 30)), flat_fraction=float(rng.uniform(0.2, 0.95))).frame(int(rng.integers(0, 40)))[0]
    if kind == 1:
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == 2:  # low texture with a few blobs: threshold fallback cells
        img = np.full((h, w), int(rng.integers(40, 200)), np.int32) + rng.integers(-2, 3, (h, w))
        for _ in range(int(rng.integers(5, 60))):
            cx, cy = int(rng.integers(0, w - 10)), int(rng.integers(0, h - 10))
            img[cy:cy + int(rng.integers(2, 12)), cx:cx + int(rng.integers(2, 12))] += int(rng.integers(-40, 40))
        return np.clip(img, 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]  # checkerboard + noise: dense corners
    p = int(rng.integers(3, 12))
    img = ((xx // p + yy // p) % 2) * int(rng.integers(30, 160)) + 50 + rng.integers(-8, 9, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


t0, n = time.time(), 0
while time.time() - t0 < budget:
    w, h = int(rng.integers(200, 1400)), int(rng.integers(160, 1000))
    nfeat = int(rng.integers(100, 3000))
    sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
    nl = int(rng.integers(1, 9))
    while min(w, h) / sf ** (nl - 1) < 70 and nl > 1:
        nl -= 1
    ini, mn = int(rng.integers(5, 40)), int(rng.integers(1, 25))
    kind = int(rng.integers(0, 5))
    cfg = dict(w=w, h=h, nfeat=nfeat, sf=sf, nl=nl, ini=ini, mn=mn, kind=kind)
    img = image(kind, w, h)
    try:
        ge = G.ORBextractor(nfeat, sf, nl, ini, mn)
        oe = O.Extractor(nfeat, sf, nl, ini, mn)
        gk, gd = ge(img)
        ok, od = oe.extract(img)
    except Exception as ex:  # both must agree on rejecting a configuration
        if "too large for the quadtree kernel" in str(ex) or "too small" in str(ex) or "aspect ratio unsupported" in str(ex):  # documented limits of the GPU path
            continue
        try:
            O.Extractor(nfeat, sf, nl, ini, mn).extract(img)
        except Exception:
            continue
        print("FAIL (GPU raised)", cfg, ex)
        sys.exit(1)
    same = len(gk) == len(ok) and all(np.array_equal(np.ascontiguousarray(gk[f]).view(np.uint32),
                                                     np.ascontiguousarray(ok[f]).view(np.uint32)) for f in FIELDS) \
        and np.array_equal(gd, od)
    if not same:
        print("FAIL", cfg, len(gk), len(ok))
        np.save("/tmp/fuzz_fail.npy", img)
        sys.exit(1)
    n += 1
print("fuzz ok: %d configurations in %.0f s" % (n, time.time() - t0))
