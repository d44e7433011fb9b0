"""FAST early-out A/B (tools/gpu_run.sh earlyout): python tools/early_out_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_map_amd import workloads

for ff in (0.5, 0.8):
    print(json.dumps(workloads.c2_fast_early_out(flat_fraction=ff)))
