"""Threading contract of the boundary: handles are not shared between threads,
distinct handles are re-entrant (the reference runs two extractors concurrently in stereo, Frame.cc:78-81), and the
stateless matcher entry points are called concurrently from Tracking, LocalMapping and LoopClosing threads -- each host
thread gets its own workspace and stream.  Several Python threads (ctypes releases the GIL inside the library) hammer
their own handles and the stateless entry points at the same time; every result equals the single-threaded one."""
import threading

import numpy as np
import pytest

import scenario

pytestmark = pytest.mark.gpu


def test_concurrent_threads_own_handles_and_stateless_matchers(gpu, oracle, stream640):
    st = stream640
    nthreads, rounds = 4, 6
    frames = [st.frame(t) for t in range(40, 40 + nthreads + 1)]
    # single-threaded references (GPU path, itself checked against the oracle elsewhere; frame 0 also against the oracle here)
    ref_ext, ref_bf, ref_proj = [], [], []
    ge = gpu.ORBextractor(1000)
    for g, _, _ in frames:
        ref_ext.append(ge(g))
    ok, od = oracle.Extractor(1000).extract(frames[0][0])
    assert ref_ext[0][0].tobytes() == ok.tobytes() and np.array_equal(ref_ext[0][1], od)
    m = gpu.ORBmatcher(0.7, True)
    sf = np.asarray(ge.GetScaleFactors(), np.float32)
    Tcw = scenario.rigid()
    rng = np.random.default_rng(3)
    probs = []
    for i in range(nthreads):
        (ka, da), (kb, db) = ref_ext[i], ref_ext[i + 1]
        ref_bf.append(m.MatchBruteForce(da, ka["angle"], db, kb["angle"]))
        # a projection problem per thread: frame i's key points as map points seen from frame i + 1
        (px, py), (ox, oy) = st.offset(40 + i), st.offset(41 + i)
        P, _ = scenario.world_points_from_prev(ka, frames[i][2], (ox - px, oy - py), st, Tcw, rng)
        mp = scenario.local_map(oracle, st, Tcw, P, da, ka["octave"], sf, rng, obs_zero_frac=0.1)
        of = scenario.make_frame(oracle, kb, db, frames[i + 1][2], st, sf)
        gf = gpu.Frame(of.kp_x, of.kp_y, of.octave, of.angle, of.u_right, of.desc, float(of.max_x), float(of.max_y), of.scale_factors)
        k0 = np.full(of.n, -1, np.int32)
        want = oracle.search_by_projection(of, mp, 3.0, 0.8, k0)
        got = gpu.ORBmatcher(0.8, True).SearchByProjection(gf, mp, 3.0, k0)
        assert got[0] == want[0] and np.array_equal(got[1], want[1])
        probs.append((gf, mp, k0))
        ref_proj.append(want)
    errors = []
    barrier = threading.Barrier(nthreads)

    def worker(i):
        try:
            ext = gpu.ORBextractor(1000)          # a handle of this thread's own
            mine = gpu.ORBmatcher(0.7, True)
            proj = gpu.ORBmatcher(0.8, True)
            tbl = gpu.MapPointTable()
            gf, mp, k0 = probs[i]
            ids = np.arange(len(mp["level"]), dtype=np.int64) * 5 + i
            tbl.upsert(ids, mp_world[i], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["obs_pos"].astype(np.int32))
            tbl.set_bad(ids[mp["bad"] != 0])
            dfr = gpu.DeviceFrame().upload(gf)
            barrier.wait()
            for r in range(rounds):
                k, d = ext(frames[i][0] if r % 2 == 0 else frames[i + 1][0])
                want = ref_ext[i] if r % 2 == 0 else ref_ext[i + 1]
                if not (k.tobytes() == want[0].tobytes() and np.array_equal(d, want[1])):
                    # what differs, and whether the same handle gets it right when asked again (a transfer or a kernel?)
                    k2_, d2_ = ext(frames[i][0] if r % 2 == 0 else frames[i + 1][0])
                    rows = np.nonzero((d != want[1]).any(1))[0] if d.shape == want[1].shape else np.zeros(0, np.int64)
                    try:  # the evidence, for offline analysis
                        import os
                        os.makedirs("gpurun_out", exist_ok=True)
                        np.savez("gpurun_out/thread_mismatch_%d_%d.npz" % (i, r), kp=k, desc=d, want_kp=want[0], want_desc=want[1],
                                 frame=np.int64(40 + i + (r % 2)))
                    except Exception:  # noqa: BLE001
                        pass
                    raise AssertionError("extraction differs (thread %d round %d): key points equal %s, %d of %d descriptor rows differ "
                                         "(first %s, differing rows per octave %s of %s, all-zero rows %d); a second call on the same handle is %s"
                                         % (i, r, k.tobytes() == want[0].tobytes(), len(rows), len(d), rows[:8].tolist(),
                                            np.bincount(want[0]["octave"][rows], minlength=8).tolist() if len(rows) else [],
                                            np.bincount(want[0]["octave"], minlength=8).tolist(), int((~d.any(1)).sum()),
                                            "right" if k2_.tobytes() == want[0].tobytes() and np.array_equal(d2_, want[1]) else "wrong too"))
                (ka, da), (kb, db) = ref_ext[i], ref_ext[i + 1]
                n, mb = mine.MatchBruteForce(da, ka["angle"], db, kb["angle"])
                assert n == ref_bf[i][0] and np.array_equal(mb, ref_bf[i][1]), "BF match differs (thread %d round %d)" % (i, r)
                n, k2 = proj.SearchByProjection(gf, mp, 3.0, k0)
                assert n == ref_proj[i][0] and np.array_equal(k2, ref_proj[i][1]), "projection differs (thread %d round %d)" % (i, r)
                n, k3 = gpu.search_local_points_table(dfr, tbl, ids, None, 0, 0, 0, 0, 0, 0, 3.0, 0.8, scratch=mp)
                assert n == ref_proj[i][0] and np.array_equal(k3, ref_proj[i][1]), "table search differs (thread %d round %d)" % (i, r)
        except Exception as ex:  # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (i, ex))
            try:
                barrier.abort()
            except Exception:
                pass

    mp_world = []
    rng = np.random.default_rng(3)
    for i in range(nthreads):  # the same world points as above (same generator sequence)
        ka = ref_ext[i][0]
        (px, py), (ox, oy) = st.offset(40 + i), st.offset(41 + i)
        P, _ = scenario.world_points_from_prev(ka, frames[i][2], (ox - px, oy - py), st, Tcw, rng)
        scenario.local_map(oracle, st, Tcw, P, ref_ext[i][1], ka["octave"], sf, rng, obs_zero_frac=0.1)  # advance the generator alike
        mp_world.append(P)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
    assert all(not t.is_alive() for t in ts)


def test_first_calls_on_fresh_handles_from_fresh_threads(gpu, stream640):
    """Four new host threads, each with a new extractor handle, make their first calls at the same moment -- 60 times over.
    A first call configures the handle (allocations, table uploads, initial values) right before it enqueues kernels;
    until round 4 the initial values were written with hipMemset, which runs on the null stream and is not ordered with
    the handle's non-blocking stream: under this load the zeroing of the blurred planes / cell counters landed after the
    kernels had written them (1 - 3 % of the runs of the test above: unwritten descriptors, wrong key points).  Also the
    regression test for recording the graph of a call instead of capturing it (tools/stress_first_calls.py is the long form)."""
    nthreads, iters = 4, 60
    frames = [stream640.frame(40 + t)[0] for t in range(nthreads + 1)]
    ge = gpu.ORBextractor(1000)
    ref = [ge(f) for f in frames]
    bad = []

    keep = []  # the handles are closed by the main thread, after the join: this test is about first calls, not teardown

    def worker(i, barrier, it):
        try:
            ext = gpu.ORBextractor(1000)
            keep.append(ext)
            barrier.wait()
            for r in range(3):  # plain launches, the graph is recorded, the graph is replayed
                k, d = ext(frames[i + (r & 1)])
                want = ref[i + (r & 1)]
                if not (k.tobytes() == want[0].tobytes() and np.array_equal(d, want[1])):
                    bad.append((it, i, r))
        except Exception as ex:  # noqa: BLE001 -- reported below
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
            t.join(120)
        assert all(not t.is_alive() for t in ts)
        while keep:
            keep.pop().close()
    assert not bad, bad[:8]


def test_short_lived_worker_threads_give_their_workspaces_back(gpu, stream640):
    """A caller that matches from short-lived worker threads: every thread's workspace (a stream, ~20 device buffers, a
    matcher handle) is released when the thread ends (csrc/workspace.h), so device memory does not grow with the number
    of threads that ever called in."""
    import torch
    st = stream640
    ge = gpu.ORBextractor(1000, max_batch=2)
    (ka, kb), (da, db) = ge.extract_batch(np.stack([st.frame(3)[0], st.frame(4)[0]]))
    want = gpu.ORBmatcher(0.7, True).MatchBruteForce(da, ka["angle"], db, kb["angle"])
    errors = []

    def worker():
        try:
            got = gpu.ORBmatcher(0.7, True).MatchBruteForce(da, ka["angle"], db, kb["angle"])
            assert got[0] == want[0] and np.array_equal(got[1], want[1])
            d = gpu.ORBmatcher.DescriptorDistance(da[:64], db[:64])
            assert len(d) == 64
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    def run(n):
        for _ in range(n):
            t = threading.Thread(target=worker)
            t.start()
            t.join(120)
            assert not t.is_alive()

    run(4)  # allocator pools, code objects, the first-use costs
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    run(40)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert not errors, errors
    # one brute-force workspace is ~0.5 MB of buffers + a matcher handle; 40 leaked ones would be tens of MB
    assert free0 - free1 < 8 << 20, "device memory shrank by %.1f MB over 40 worker threads" % ((free0 - free1) / 2**20)
