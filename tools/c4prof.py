"""C4 only (dense-map insert), for rocprofv3 iterations: python tools/c4prof.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_map_amd import workloads as W

r = W.c4()
r.pop("per_keyframe")
print(json.dumps({k: v for k, v in r.items() if k not in ("workload",)}))
