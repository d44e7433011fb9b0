import sys, json
sys.path.insert(0, '/root/repo')
from orb_slam2_map_amd import workloads as W
import numpy as np
orig = W.c3
# monkeypatch nprev through source edit: simplest is to call internals -> re-exec with different nprev
import inspect, types
src = inspect.getsource(W.c3)
for nprev in (1, 2, 3, 5):
    ns = dict(W.__dict__)
    exec(src.replace("nprev, t_cur = 5, 12", "nprev, t_cur = %d, 12" % nprev), ns)
    r = ns['c3'](reps=20)
    print(nprev, r['map_points'], 'search_ms', round(r['search_ms'], 4), 'sweeps', r['claim_sweeps'], 'matches', r['matches'], flush=True)
