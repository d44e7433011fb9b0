"""First calls on fresh handles from fresh host threads, many times in one process: the situation in which round 4 found
first calls returning unwritten descriptors / wrong key points (1 - 3 % of tests/test_gpu_threads.py's runs; DESIGN.md
section 9.12: configure()'s hipMemset ran on the null stream, behind the kernels of the handle's non-blocking stream).
Every result is compared with the single-threaded one.
    python tools/stress_first_calls.py [iterations] [threads]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
st = Stream(640, 480, 1234)
frames = [st.frame(40 + t)[0] for t in range(nthreads + 1)]
ge = G.ORBextractor(1000)
ref = [ge(f) for f in frames]
bad = []


def worker(i, barrier, it):
    try:
        ext = G.ORBextractor(1000)
        m = G.ORBmatcher(0.7, True)
        barrier.wait()
        for r in range(3):  # plain launches, the graph is recorded, the graph is replayed
            k, d = ext(frames[i + (r & 1)])
            want = ref[i + (r & 1)]
            if not (k.tobytes() == want[0].tobytes() and np.array_equal(d, want[1])):
                bad.append((it, i, r))
            m.MatchBruteForce(ref[i][1], ref[i][0]["angle"], ref[i + 1][1], ref[i + 1][0]["angle"])
    except Exception as ex:  # noqa: BLE001 -- counted as a failure
        bad.append((it, i, repr(ex)[:200]))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


for it in range(iters):
    barrier = threading.Barrier(nthreads)
    ts = [threading.Thread(target=worker, args=(i, barrier, it)) for i in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
print("stress_first_calls: %d iterations x %d threads, wrong results: %d %s" % (iters, nthreads, len(bad), bad[:6]))
sys.exit(1 if bad else 0)
