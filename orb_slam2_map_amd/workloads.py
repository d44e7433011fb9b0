"""Secondary workloads of BASELINE.json (C3, C4) measured through the C ABI -- used by bench.py (`secondary`) and
tools/bench_extra.py.  Inputs are synthetic and built with numpy only (no oracle, no reference); parity of the same
entry points is the job of tests/test_gpu_matcher_proj.py and tests/test_gpu_cloud.py.

Durations are HIP-event times on the stream the kernels run on; bytes are SURVEY.md 8d's algorithmic bytes.
"""
import time

import numpy as np

from . import lib as G
from .synth import Stream

HBM_PEAK_GBS = 8000.0


def rigid(rx=0.01, ry=-0.02, rz=0.015, t=(0.03, -0.02, 0.05)):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T.astype(np.float32)


def _world_points(kps, depth, shift, st, Tcw, rng, jitter=0.6):
    """World positions for a previous frame's key points so that they project, in the current frame, onto the
    image content they came from (the stream is a translating window of one canvas)."""
    fx, fy, cx, cy = float(st.fx), float(st.fy), float(st.cx), float(st.cy)
    u = kps["x"].astype(np.float64) - shift[0] + rng.normal(0, jitter, len(kps))
    v = kps["y"].astype(np.float64) - shift[1] + rng.normal(0, jitter, len(kps))
    d = depth[kps["y"].astype(np.int64), kps["x"].astype(np.int64)].astype(np.float64)
    d = np.where(d > 0, d, 2.0)
    Pc = np.stack([(u - cx) * d / fx, (v - cy) * d / fy, d], 1)
    T = Tcw.astype(np.float64)
    return ((Pc - T[:3, 3]) @ T[:3, :3]).astype(np.float32)


def local_map_table(world_pos, desc, octave, scale_factors, Tcw, rng):
    """MapPoint fields Frame::isInFrustum reads (normal, invariance distances as UpdateNormalAndDepth sets them)."""
    sf = np.asarray(scale_factors, np.float32)
    T = Tcw.astype(np.float64)
    Ow = -T[:3, :3].T @ T[:3, 3]
    P = world_pos.astype(np.float64)
    dist = np.linalg.norm(P - Ow, axis=1)
    normal = (P - Ow) / np.maximum(dist, 1e-9)[:, None]
    max_d = (dist * sf[octave]).astype(np.float32)
    min_d = (max_d / sf[-1]).astype(np.float32)
    m = len(P)
    return {"world_pos": np.ascontiguousarray(world_pos, np.float32), "normal": normal.astype(np.float32),
            "min_dist": min_d, "max_dist": max_d, "desc": np.ascontiguousarray(desc, np.uint8),
            "skip": (rng.random(m) < 0.02).astype(np.uint8), "obs_pos": np.ones(m, np.uint8)}


def c3(reps=30, device_id=0):
    """C3: 1280x960, 2000 features, extract + SearchByProjection against ~10 k local map points, device resident
    (Tracking::SearchLocalPoints: isInFrustum + ORBmatcher::SearchByProjection(F, vpMapPoints, th))."""
    import torch
    W, H, NF = 1280, 960, 2000
    rng = np.random.default_rng(5678)
    st = Stream(W, H, 1234)
    nprev, t_cur = 5, 12
    ts = [t_cur - 1 - i for i in range(nprev)] + [t_cur]
    frames = [st.frame(t) for t in ts]
    ext = G.ORBextractor(NF, max_batch=nprev + 1, device_id=device_id)
    ks, ds = ext.extract_batch(np.stack([f[0] for f in frames]))
    sf = ext.GetScaleFactors()
    Tcw = rigid()
    ox, oy = st.offset(t_cur)
    wp, dsc, octv = [], [], []
    for i, t in enumerate(ts[:-1]):
        px, py = st.offset(t)
        wp.append(_world_points(ks[i], frames[i][2], (ox - px, oy - py), st, Tcw, rng))
        dsc.append(ds[i]), octv.append(ks[i]["octave"])
    wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
    tab = local_map_table(wp, dsc, octv, sf, Tcw, rng)
    M = len(wp)

    dev = "cuda:%d" % device_id
    img = torch.from_numpy(frames[-1][0][None]).to(dev)
    depth = torch.from_numpy(frames[-1][2][None]).to(dev)
    g1 = G.ORBextractor(NF, max_batch=1, device_id=device_id)
    cap = g1.max_keypoints(W, H)
    kps = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
    nout = torch.zeros(1, dtype=torch.int32, device=dev)
    ur, dz = (torch.zeros((1, cap), dtype=torch.float32, device=dev) for _ in range(2))
    cs = torch.zeros((1, 64 * 48 + 1), dtype=torch.int32, device=dev)
    items = torch.zeros((1, cap), dtype=torch.int32, device=dev)
    dtab = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in tab.items()}
    sfa = np.asarray(sf, np.float32)
    fv = G.DeviceFrameView()
    fv.cap, fv.n, fv.kps, fv.desc, fv.u_right = cap, nout.data_ptr(), kps.data_ptr(), desc.data_ptr(), ur.data_ptr()
    fv.cell_start, fv.cell_items, fv.nlevels, fv.scale_factors = cs.data_ptr(), items.data_ptr(), len(sfa), sfa.ctypes.data
    fv.min_x, fv.max_x, fv.min_y, fv.max_y = 0.0, float(W), 0.0, float(H)
    tb = G.DeviceMapPointTable()
    tb.m = M
    for k in dtab:
        setattr(tb, k, dtab[k].data_ptr())
    k2m = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    log_sf = float(np.log(np.float32(sfa[1])))
    cam = G.make_camera(float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf), W, H)
    camv = (float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf))

    def extract_glue():
        g1.extract_batch_device(img.data_ptr(), 1, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
        G.frame_glue_batch_device(1, cap, kps.data_ptr(), nout.data_ptr(), depth.data_ptr(), W, W * H, cam, None,
                                  ur.data_ptr(), dz.data_ptr(), cs.data_ptr(), items.data_ptr(), s, device_id)

    def search():
        G.search_local_points_device(fv, tb, Tcw, *camv, log_sf, 3.0, 0.8, k2m.data_ptr(), counts.data_ptr(), None,
                                     stream=s, device_id=device_id)

    def timed(fn, pre=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(reps):
            if pre:
                pre()
            e0.record(stream)
            fn()
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))  # a single stalled repetition (seen once: 38 ms among 30 of 0.33 ms) must not set the figure

    extract_glue()
    k2m.fill_(-1)
    # one untimed call with the tracking scratch attached: mbTrackInView of every point = the real in-view count
    in_view = torch.zeros(M, dtype=torch.uint8, device=dev)
    trk = G.TrackScratch()
    trk.in_view = in_view.data_ptr()
    G.search_local_points_device(fv, tb, Tcw, *camv, log_sf, 3.0, 0.8, k2m.data_ptr(), counts.data_ptr(), trk, stream=s,
                                 device_id=device_id)
    torch.cuda.synchronize()
    n_view, n_bad_level = int(in_view.sum()), int(counts[1])
    k2m.fill_(-1)
    search()
    torch.cuda.synchronize()
    n_match = int(counts[0])
    N = int(nout[0])
    ms_search = timed(search, pre=lambda: k2m.fill_(-1))
    ms_extract = timed(extract_glue)
    ms_chain = timed(lambda: (extract_glue(), search()), pre=lambda: k2m.fill_(-1))
    sweeps, rewalked = G.projection_last_sweeps()
    # M2 reads M*60 + N*48 + the grid (20 292 B), writes (M + N)*4
    alg_m2 = M * 60 + N * 48 + 20292 + (M + N) * 4
    px = [1228800, 853600, 592963, 411996, 285671, 198404, 138138, 95676]
    alg_ext = sum(px[:-1]) + sum(px[1:]) + sum(px) + 2 * sum(px) + N * 1321  # 8d: 20 343 764 B at N = 2000
    ach = alg_m2 / (ms_search * 1e-3) / 1e9
    return {"workload": "C3: synthetic 1280x960, 2000 features, extract + isInFrustum + SearchByProjection(th=3) of %d "
                        "local map points (%d in view), device resident, one frame at a time" % (M, n_view),
            "keypoints": N, "map_points": M, "in_view": n_view, "levels_out_of_range": n_bad_level, "matches": n_match,
            "claim_sweeps": sweeps, "rewalked_rows": rewalked,
            "search_ms": ms_search, "extract_glue_ms": ms_extract, "extract_glue_search_ms": ms_chain,
            "frames_per_s": 1e3 / ms_chain,
            "roofline": {"bound": "hbm", "kernel": "k_frustum_queries + k_proj_lists + k_proj_resolve", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes": alg_m2, "ms_per_launch": ms_search,
                         "note": "one 10 k-point query set per call: gather-latency / launch bound, not a stream"},
            "extract_roofline": {"bound": "hbm", "achieved": alg_ext / (ms_extract * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": alg_ext / (ms_extract * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes": alg_ext, "note": "single frame: 14 dependent launches"}}


class C3Batch:
    """C3 in throughput mode: n_seq independent sequences (frames shard by sequence), one 1280x960 frame each per step:
    batched extraction + frame glue + orbgpu_search_local_points_batch_device (= per frame Tracking::SearchLocalPoints,
    Tracking.cc:1447-1497 -> ORBmatcher::SearchByProjection(F, vpMapPoints, th), ORBmatcher.cc:45-129).  Every sequence
    has its own copy of the frame, of the ~10 k-point map table and of all outputs (distinct addresses); the content is
    the C3 scenario of the stream with seed `seed` (one seed per rank in bench.py --workload c3_batch)."""

    W, H, NF = 1280, 960, 2000
    PX = [1228800, 853600, 592963, 411996, 285671, 198404, 138138, 95676]

    def __init__(self, n_seq=128, device_id=0, seed=1234):
        import torch
        W, H, NF = self.W, self.H, self.NF
        rng = np.random.default_rng(5678)
        st = Stream(W, H, seed)
        nprev, t_cur = 5, 12
        ts = [t_cur - 1 - i for i in range(nprev)] + [t_cur]
        frames = [st.frame(t) for t in ts]
        self.host_frames = [f[0] for f in frames]
        self.ext = ext = G.ORBextractor(NF, max_batch=n_seq, device_id=device_id)
        ks, ds = ext.extract_batch(np.stack([f[0] for f in frames]))
        sf = ext.GetScaleFactors()
        self.Tcw = Tcw = rigid()
        ox, oy = st.offset(t_cur)
        wp, dsc, octv = [], [], []
        for i, t in enumerate(ts[:-1]):
            px, py = st.offset(t)
            wp.append(_world_points(ks[i], frames[i][2], (ox - px, oy - py), st, Tcw, rng))
            dsc.append(ds[i]), octv.append(ks[i]["octave"])
        wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
        self.table = tab = local_map_table(wp, dsc, octv, sf, Tcw, rng)
        self.M = M = len(wp)
        self.st = st
        dev = "cuda:%d" % device_id
        self.B = B = n_seq
        self.device_id = device_id
        self.img = torch.from_numpy(frames[-1][0]).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
        self.depth = torch.from_numpy(frames[-1][2]).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
        self.cap = cap = ext.max_keypoints(W, H)
        self.kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        self.nout = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ur, self.dz = (torch.zeros((B, cap), dtype=torch.float32, device=dev) for _ in range(2))
        self.cs = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device=dev)
        self.items = torch.zeros((B, cap), dtype=torch.int32, device=dev)
        self.dtab = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev).unsqueeze(0).repeat(*([B] + [1] * v.ndim)).contiguous()
                     for k, v in tab.items()}
        self.k2m = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
        self.counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        self.sfa = sfa = np.asarray(sf, np.float32)
        self.log_sf = log_sf = float(np.log(np.float32(sfa[1])))
        self.cam = G.make_camera(float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf), W, H)
        self.problems, self.keep = [], []
        for b in range(B):
            fv = G.DeviceFrameView()
            fv.cap, fv.n, fv.kps, fv.desc = cap, self.nout[b:].data_ptr(), self.kps[b].data_ptr(), self.desc[b].data_ptr()
            fv.u_right, fv.cell_start, fv.cell_items = self.ur[b].data_ptr(), self.cs[b].data_ptr(), self.items[b].data_ptr()
            fv.nlevels, fv.scale_factors = len(sfa), sfa.ctypes.data
            fv.min_x, fv.max_x, fv.min_y, fv.max_y = 0.0, float(W), 0.0, float(H)
            tb = G.DeviceMapPointTable()
            tb.m = M
            for k in self.dtab:
                setattr(tb, k, self.dtab[k][b].data_ptr())
            self.keep.append((fv, tb))
            self.problems.append({"frame": fv, "table": tb, "Tcw": Tcw, "fx": float(st.fx), "fy": float(st.fy),
                                  "cx": float(st.cx), "cy": float(st.cy), "mbf": float(st.bf), "log_sf": log_sf,
                                  "d_kp_to_mp": self.k2m[b].data_ptr(), "d_counts": self.counts[b].data_ptr()})
        self.stream = torch.cuda.current_stream()
        self.s = self.stream.cuda_stream

    def extract_glue(self):
        B, W, H, cap, s = self.B, self.W, self.H, self.cap, self.s
        self.ext.extract_batch_device(self.img.data_ptr(), B, W, H, W, W * H, self.kps.data_ptr(), self.desc.data_ptr(), cap,
                                      self.nout.data_ptr(), s)
        G.frame_glue_batch_device(B, cap, self.kps.data_ptr(), self.nout.data_ptr(), self.depth.data_ptr(), W, W * H, self.cam,
                                  None, self.ur.data_ptr(), self.dz.data_ptr(), self.cs.data_ptr(), self.items.data_ptr(), s,
                                  self.device_id)

    def search(self):
        G.search_local_points_batch_device(self.problems, 0.5, 3.0, 0.8, stream=self.s, device_id=self.device_id)

    def step(self):
        """One frame of every sequence: nothing is associated yet (a fresh Frame), extract + glue, then the search."""
        self.k2m.fill_(-1)
        self.extract_glue()
        self.search()

    def algorithmic_bytes(self, n_kp):
        """(search, extraction) bytes per step, SURVEY.md 8d: M2 reads M*60 + N*48 + the grid (20 292 B), writes (M + N)*4;
        extraction 20 343 764 B per frame at N = 2000."""
        px = self.PX
        alg_m2 = (self.M * 60 + n_kp * 48 + 20292 + (self.M + n_kp) * 4) * self.B
        alg_ext = (sum(px[:-1]) + sum(px[1:]) + sum(px) + 2 * sum(px) + n_kp * 1321) * self.B
        return alg_m2, alg_ext

    def stage_bytes(self, stage, n_kp, n_cand):
        """SURVEY.md 8d per stage for one 1280x960 frame."""
        px = self.PX
        tot = sum(px)
        return {"pyramid": sum(px[:-1]) + sum(px[1:]), "fast": tot + 4 * n_cand, "blur": 2 * tot,
                "orient": n_kp * (749 + 28 + 16), "describe": n_kp * (512 + 32),
                "quadtree": 4 * n_cand * 2 + 4 * n_kp}[stage]


def c3_batch(n_seq=128, reps=10, device_id=0):
    """bench.py `secondary.c3_batch`: the C3Batch step timed piecewise with HIP events on one GPU."""
    import torch
    wl = C3Batch(n_seq, device_id)
    B, M, stream = wl.B, wl.M, wl.stream

    def timed(fn, pre=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(reps):
            if pre:
                pre()
            e0.record(stream)
            fn()
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))  # a single stalled repetition (seen once: 38 ms among 30 of 0.33 ms) must not set the figure

    wl.step()
    torch.cuda.synchronize()
    cnt = wl.counts.cpu().numpy()
    assert (cnt[:, 0] == cnt[0, 0]).all() and cnt[0, 0] > 0, "identical problems must give identical match counts"
    N = int(wl.nout[0])
    ms_search = timed(wl.search, pre=lambda: wl.k2m.fill_(-1))
    wl.ext.set_profiling(True)
    ms_extract = timed(wl.extract_glue)
    stages = wl.ext.stage_times()
    wl.ext.set_profiling(False)
    ms_chain = timed(lambda: (wl.extract_glue(), wl.search()), pre=lambda: wl.k2m.fill_(-1))
    alg_m2, alg_ext = wl.algorithmic_bytes(N)
    ach_s = alg_m2 / (ms_search * 1e-3) / 1e9
    ach_e = alg_ext / (ms_extract * 1e-3) / 1e9
    return {"workload": "C3 throughput mode: %d independent sequences, one synthetic 1280x960 frame each per step, 2000 "
                        "features, batched extract + glue + isInFrustum + SearchByProjection(th=3) of %d map points per "
                        "sequence, device resident" % (B, M),
            "sequences": B, "keypoints": N, "map_points": M, "matches_per_frame": int(cnt[0, 0]),
            "search_ms": ms_search, "extract_glue_ms": ms_extract, "step_ms": ms_chain, "frames_per_s": B * 1e3 / ms_chain,
            "extract_stage_ms": {k: round(v, 4) for k, v in stages.items()},
            "roofline": {"bound": "hbm", "kernel": "extraction stages at 1280x960 (20.3 MB/frame)", "achieved": ach_e,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_e / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes": alg_ext, "ms_per_launch": ms_extract},
            "search_roofline": {"bound": "hbm", "kernel": "k_frustum_queries_batch + k_proj_lists_batch + k_proj_resolve_batch",
                                "achieved": ach_s, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_s / HBM_PEAK_GBS,
                                "algorithmic_bytes": alg_m2, "ms_per_launch": ms_search,
                                "note": "one workgroup per sequence runs the claim fixpoint: latency bound per problem, "
                                        "parallel across sequences"}}


def c4(nkf=24, device_id=0, leaf=0.01):
    """C4: 640x480 key frames: extract + BF match against the previous key frame + dense-map insert (stride-3
    back-projection, transform, voxel filter of map + new points at 0.01 m), images resident in HBM.
    Two passes over the same key frames: (a) serial -- one host thread does extract + match, waits, inserts, waits (the
    round-1..3 figure); (b) as the reference runs it -- PointCloudMapping has a thread of its own (PointCloudMap.cc:53,
    `viewer`): the tracking thread hands a key frame over (`insertKeyFrame`, :69-76) and goes on with the next frame while
    the map thread inserts on the cloud handle's stream."""
    import queue
    import threading
    import torch
    W, H, NF = 640, 480, 1000
    dev = "cuda:%d" % device_id
    st = Stream(W, H, 1234)
    frames = [st.frame(7 * i) for i in range(nkf)]
    poses = [rigid(0.004 * i, -0.006 * i, 0.002 * i, (0.04 * i, 0.01 * i, 0.015 * i)) for i in range(nkf)]
    cam = (float(st.fx), float(st.fy), float(st.cx), float(st.cy))
    gray = torch.from_numpy(np.stack([f[0] for f in frames])).to(dev)
    rgb = torch.from_numpy(np.stack([f[1] for f in frames])).to(dev)
    depth = torch.from_numpy(np.stack([f[2] for f in frames])).to(dev)
    n_new = [int(np.count_nonzero(~((f[2][::3, ::3].astype(np.float64) < 0.01) | (f[2][::3, ::3] > 10)))) for f in frames]
    ext = G.ORBextractor(NF, max_batch=1, device_id=device_id)
    cap = ext.max_keypoints(W, H)
    matcher = G.BatchMatcher(1, cap, device_id=device_id)
    kps = torch.zeros((2, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
    nout = torch.zeros(2, dtype=torch.int32, device=dev)
    mb = torch.zeros((1, cap), dtype=torch.int32, device=dev)
    nm = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    KP, DS = cap * 28, cap * 32
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def track(i):  # Tracking's share of a key frame: extraction + match against the previous one, then wait for it
        cur, prv = i & 1, (i & 1) ^ 1
        e0.record(stream)
        ext.extract_batch_device(gray[i].data_ptr(), 1, W, H, W, W * H, kps.data_ptr() + cur * KP,
                                 desc.data_ptr() + cur * DS, cap, nout.data_ptr() + 4 * cur, s)
        matcher.match(1, cap, desc.data_ptr() + prv * DS, kps.data_ptr() + prv * KP + 12, None, nout.data_ptr() + 4 * prv,
                      desc.data_ptr() + cur * DS, kps.data_ptr() + cur * KP + 12, nout.data_ptr() + 4 * cur, 28, 50, 0.7,
                      True, mb.data_ptr(), nm.data_ptr(), s)
        e1.record(stream)

    # (a) serial
    cloud = G.PointCloudMapping(leaf, device_id)
    cloud.set_profiling(True)
    rows = []
    for i in range(nkf):
        k_before = cloud.size()
        t0 = time.perf_counter()
        track(i)
        cloud.insertKeyFrameDevice(depth[i].data_ptr(), W, rgb[i].data_ptr(), W * 3, W, H, *cam, poses[i])
        e1.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ins = cloud.last_insert_ms()
        v = cloud.size()
        alg = ((H + 2) // 3) * W * (4 + 3) + n_new[i] * 16 + k_before * 16 + v * 16  # 8d: P1 + P3
        rows.append({"map_before": k_before, "new_points": n_new[i], "map_after": v, "insert_ms": ins,
                     "extract_match_ms": e0.elapsed_time(e1), "wall_ms": wall, "path": cloud.last_path(),
                     "algorithmic_bytes": alg})
    final_serial = cloud.size()
    cloud.close()

    # (b) two host threads, as in the reference
    cloud = G.PointCloudMapping(leaf, device_id)
    q = queue.Queue()
    err = []

    def viewer():
        torch.cuda.set_device(device_id)
        while True:
            i = q.get()
            if i is None:
                return
            try:
                cloud.insertKeyFrameDevice(depth[i].data_ptr(), W, rgb[i].data_ptr(), W * 3, W, H, *cam, poses[i])
            except Exception as ex:  # noqa: BLE001 -- reported by the caller
                err.append(repr(ex))

    th = threading.Thread(target=viewer)
    th.start()
    stamps = []
    for i in range(nkf):
        track(i)
        e1.synchronize()
        q.put(i)
        stamps.append(time.perf_counter())
    q.put(None)
    th.join()
    t_end = time.perf_counter()
    if err:
        raise RuntimeError("map thread: " + err[0])
    assert cloud.size() == final_serial, "the threaded run built another map"
    cloud.close()
    half = nkf // 2
    wall_threaded = (t_end - stamps[half - 1]) / (nkf - half) * 1e3  # second half, map thread's tail included

    steady = rows[nkf // 2:]
    ins_ms = float(np.mean([r["insert_ms"] for r in steady]))
    alg = float(np.mean([r["algorithmic_bytes"] for r in steady]))
    ach = alg / (ins_ms * 1e-3) / 1e9
    last = rows[-1]
    serial_wall = float(np.mean([r["wall_ms"] for r in steady]))
    return {"workload": "C4: 640x480 key frames, extract + BF match vs previous key frame + dense-map insert (stride-3 "
                        "back-projection, SE3 transform, voxel filter of map + new at %.3g m), %d key frames, device "
                        "resident" % (leaf, nkf),
            "keyframes": nkf, "final_map_points": last["map_after"], "paths": sorted(set(r["path"] for r in rows)),
            "insert_ms_first": rows[0]["insert_ms"], "insert_ms_last": last["insert_ms"], "insert_ms_mean_2nd_half": ins_ms,
            "extract_match_ms_mean": float(np.mean([r["extract_match_ms"] for r in steady])),
            "keyframe_wall_ms_mean": wall_threaded,
            "keyframes_per_s": 1e3 / wall_threaded,
            "schedule": "two host threads as in the reference (PointCloudMap.cc:53): tracking = extract + match on its stream, the "
                        "map thread inserts on the cloud handle's stream; second half of the key frames, the map thread's tail included",
            "serial": {"keyframe_wall_ms_mean": serial_wall, "keyframes_per_s": 1e3 / serial_wall,
                       "note": "one host thread: extract + match, insert, wait (the figure of rounds 1 - 3)"},
            "roofline": {"bound": "hbm", "kernel": "k_bp + k_vox_keys + 4 k_sort_pass + k_merge_new + k_merge_old",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": alg, "ms_per_launch": ins_ms,
                         "note": "8 dependent launches per key frame over <= 34 k new points + one pass over the map"},
            "per_keyframe": rows}


def c2_fast_early_out(batch=256, reps=8, device_id=0, flat_fraction=0.5):
    """The FAST stage with and without its exact wave-level early-out (orbgpu_extractor_set_fast_early_out) on two 640x480
    streams: the benchmark stream (textured everywhere: the option's worst case) and a stream with TUM-desk-like flat
    regions (Stream(..., flat_fraction)).  Extraction only, resident batch, HIP events of the library per stage."""
    import torch
    W, H, NF = 640, 480, 1000
    dev = "cuda:%d" % device_id
    out = {"batch": batch, "flat_fraction": flat_fraction,
           "what": "ms per %d-frame launch of k_fast_detect / of the whole extraction, option off and on; key points are "
                   "bit-identical either way (tests/test_gpu_extractor.py)" % batch}
    ext = G.ORBextractor(NF, max_batch=batch, device_id=device_id)
    cap = ext.max_keypoints(W, H)
    kps = torch.zeros((batch, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((batch, cap, 32), dtype=torch.uint8, device=dev)
    nout = torch.zeros(batch, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for name, ff in (("benchmark_stream", 0.0), ("flat_region_stream", flat_fraction)):
        st = Stream(W, H, 1234, flat_fraction=ff)
        imgs = torch.from_numpy(np.stack([st.frame(t)[0] for t in range(batch)])).to(dev)
        row = {}
        for on in (False, True):
            ext.set_fast_early_out(on)
            for i in range(reps + 2):
                ext.set_profiling(i >= 2)
                ext.extract_batch_device(imgs.data_ptr(), batch, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap,
                                         nout.data_ptr(), s)
            torch.cuda.synchronize()
            stg = ext.stage_times()
            ext.set_profiling(False)
            row["early_out_on" if on else "early_out_off"] = {"fast_ms": round(stg["fast"], 4), "extract_ms": round(sum(stg.values()), 4),
                                                               "extract_frames_per_s": round(batch / (sum(stg.values()) * 1e-3), 0),
                                                               "keypoints_per_frame": float(nout.float().mean())}
        row["fast_speedup_with_option"] = round(row["early_out_off"]["fast_ms"] / row["early_out_on"]["fast_ms"], 3)
        out[name] = row
    ext.set_fast_early_out(False)
    return out
