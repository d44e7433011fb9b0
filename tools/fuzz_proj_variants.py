"""Random-parameter sweep of the other projection matchers (last frame host / device resident, key frame, Sim3, local
points device resident, crowded windows): the scenario builders of tests/test_gpu_matcher_proj.py.
    python tools/fuzz_proj_variants.py [seconds] [seed]"""
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
t0 = time.time()
count = {}
MOTIONS = ["none", "forward", "backward"]
while time.time() - t0 < budget:
    k = int(rng.integers(0, 6))
    try:
        if k == 0:
            args = (float(rng.choice([3.0, 7.0, 15.0, 30.0])), bool(rng.integers(0, 2)), float(rng.choice([0.0, 0.2, 0.4])),
                    str(rng.choice(MOTIONS)))
            T.test_search_by_projection_last_frame(G, O, *args)
        elif k == 1:
            args = (float(rng.choice([3.0, 7.0, 15.0, 30.0])), bool(rng.integers(0, 2)), float(rng.choice([0.0, 0.2, 0.4])),
                    str(rng.choice(MOTIONS)))
            T.test_search_by_projection_last_frame_device_resident(G, O, *args)
        elif k == 2:
            args = (float(rng.choice([2.0, 3.0, 10.0])), int(rng.choice([50, 64, 100])), float(rng.choice([0.0, 0.3, 0.6])))
            T.test_search_by_projection_keyframe(G, O, *args)
        elif k == 3:
            args = (int(rng.choice([3, 4, 10])), float(rng.choice([0.6, 1.0, 1.3, 1.7])), int(rng.choice([0, 60, 150])))
            T.test_search_by_projection_sim3_loop_closing(G, O, *args)
        elif k == 4:
            args = (float(rng.choice([1.0, 3.0, 5.0])), float(rng.choice([0.0, 0.3])))
            T.test_search_local_points_device_resident(G, O, *args)
        else:
            th = float(rng.choice([1.5, 3.0, 6.0]))
            args = (th, th >= 3.0)
            T.test_search_by_projection_crowded_windows(G, O, *args)
    except AssertionError as ex:
        print("FAIL", k, args, str(ex)[:300])
        sys.exit(1)
    count[k] = count.get(k, 0) + 1
print("fuzz ok:", count, "in %.0f s" % (time.time() - t0))
