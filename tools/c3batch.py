"""C3 throughput mode only: python tools/c3batch.py [n_seq]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_map_amd import workloads as W

print(json.dumps(W.c3_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 64)))
