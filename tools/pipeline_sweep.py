"""Experiment: C2 step (extract 512 frames + BF match of consecutive frames) with the extraction as one B=512 call or as
P sub-batches on P streams, part k starting when part k-1 has passed `after` (ring over steps).  Matcher of step i on its
own stream next to extraction i+1; two output sets."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
args = ap.parse_args()
W, H, B, POOL = 640, 480, 512, 1024
st = Stream(W, H, 1234)
fr = np.stack([st.frame(t)[0] for t in range(32)])
frames = torch.from_numpy(np.concatenate([fr] * (POOL // 32))).cuda()


def run_c(P, with_match=True):
    """the same with orbgpu_pipeline (stagger after the pyramid)"""
    pl = G.ExtractorPipeline(1000, max_batch=B, parts=P)
    cap = pl.max_keypoints(W, H)
    KP, DS = cap * 28, cap * 32
    kps = [torch.zeros((B + 1, cap, 7), dtype=torch.float32, device="cuda") for _ in range(2)]
    desc = [torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda") for _ in range(2)]
    nout = [torch.zeros(B + 1, dtype=torch.int32, device="cuda") for _ in range(2)]
    matcher = G.BatchMatcher(B, cap)
    mb = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    s_match = torch.cuda.Stream()
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_match = [torch.cuda.Event() for _ in range(2)]
    for e in ev_done + ev_match:
        e.record(s_match)

    def step(i):
        s = i % 2
        pl.extract_batch_device(frames[(i % 2) * B:].data_ptr(), B, W, H, W, W * H, kps[s].data_ptr() + KP, desc[s].data_ptr() + DS, cap,
                                nout[s].data_ptr() + 4, ev_match[s].cuda_event, ev_done[s].cuda_event)
        if with_match:
            with torch.cuda.stream(s_match):
                s_match.wait_event(ev_done[s])
                matcher.match(B, cap, desc[s].data_ptr(), kps[s].data_ptr() + 12, None, nout[s].data_ptr(), desc[s].data_ptr() + DS,
                              kps[s].data_ptr() + KP + 12, nout[s].data_ptr() + 4, 28, 50, 0.7, True, mb.data_ptr(), nm.data_ptr(), s_match.cuda_stream)
                kps[1 - s][0].copy_(kps[s][B], non_blocking=True)
                desc[1 - s][0].copy_(desc[s][B], non_blocking=True)
                nout[1 - s][0:1].copy_(nout[s][B:B + 1], non_blocking=True)
                ev_match[s].record(s_match)
    return step, lambda: (int(nout[0][1:].sum()), int(nm.sum()))


def run_fused(P, after):
    """every part extracts AND matches its sub-batch on its own stream (the first frame of part k against the last frame of
    part k-1: one event); parts staggered as before"""
    exts = [G.ORBextractor(1000, max_batch=B // P) for _ in range(P)]
    cap = exts[0].max_keypoints(W, H)
    KP, DS = cap * 28, cap * 32
    n = B // P
    NS = 3
    kps = [torch.zeros((B + 1, cap, 7), dtype=torch.float32, device="cuda") for _ in range(NS)]
    desc = [torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda") for _ in range(NS)]
    nout = [torch.zeros(B + 1, dtype=torch.int32, device="cuda") for _ in range(NS)]
    matchers = [G.BatchMatcher(n, cap) for _ in range(P)]
    mb = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(P)]
    ev_stage = [torch.cuda.Event() for _ in range(P)]
    ev_ext = [[torch.cuda.Event() for _ in range(P)] for _ in range(NS)]
    ev_all = [[torch.cuda.Event() for _ in range(P)] for _ in range(NS)]
    for k in range(P):
        ev_stage[k].record(streams[k])
        exts[k].set_stage_signal(after, ev_stage[k].cuda_event)

    def step(i):
        s, ps = i % NS, (i - 1) % NS
        for k in range(P):
            st = streams[k]
            if not (i == 0 and k == 0):
                st.wait_event(ev_stage[(k - 1) % P])
            for kk in range(P):
                st.wait_event(ev_all[s][kk])  # set s was last used NS steps ago: all of its readers are done
            exts[k].extract_batch_device(frames[(i % 2) * B + k * n:].data_ptr(), n, W, H, W, W * H, kps[s].data_ptr() + KP * (1 + k * n),
                                         desc[s].data_ptr() + DS * (1 + k * n), cap, nout[s].data_ptr() + 4 * (1 + k * n), st.cuda_stream)
            ev_ext[s][k].record(st)
            with torch.cuda.stream(st):
                if k == 0:
                    if i > 0:
                        st.wait_event(ev_ext[ps][P - 1])
                        kps[s][0].copy_(kps[ps][B], non_blocking=True)
                        desc[s][0].copy_(desc[ps][B], non_blocking=True)
                        nout[s][0:1].copy_(nout[ps][B:B + 1], non_blocking=True)
                else:
                    st.wait_event(ev_ext[s][k - 1])
                o = k * n
                matchers[k].match(n, cap, desc[s].data_ptr() + DS * o, kps[s].data_ptr() + KP * o + 12, None, nout[s].data_ptr() + 4 * o,
                                  desc[s].data_ptr() + DS * (o + 1), kps[s].data_ptr() + KP * (o + 1) + 12, nout[s].data_ptr() + 4 * (o + 1), 28, 50, 0.7, True,
                                  mb.data_ptr() + 4 * cap * o, nm.data_ptr() + 4 * o, st.cuda_stream)
                ev_all[s][k].record(st)
    return step, lambda: (int(nout[0][1:].sum()), int(nm.sum()))


def run(P, after, with_match=True):
    if after == "C":
        return run_c(P, with_match)
    if after.startswith("fused-"):
        return run_fused(P, after[6:])
    exts = [G.ORBextractor(1000, max_batch=B // P) for _ in range(P)]
    cap = exts[0].max_keypoints(W, H)
    KP, DS = cap * 28, cap * 32
    kps = [torch.zeros((B + 1, cap, 7), dtype=torch.float32, device="cuda") for _ in range(2)]
    desc = [torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda") for _ in range(2)]
    nout = [torch.zeros(B + 1, dtype=torch.int32, device="cuda") for _ in range(2)]
    matcher = G.BatchMatcher(B, cap)
    mb = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(P)]
    s_match = torch.cuda.Stream()
    ev_stage = [torch.cuda.Event() for _ in range(P)]
    ev_done = [[torch.cuda.Event() for _ in range(P)] for _ in range(2)]
    ev_match = [torch.cuda.Event() for _ in range(2)]
    if P > 1 and after != "free":
        for k in range(P):
            ev_stage[k].record(streams[k])
            exts[k].set_stage_signal(after, ev_stage[k].cuda_event)
    n = B // P

    def step(i):
        s = i % 2
        for k in range(P):
            if P > 1 and after != "free" and not (i == 0 and k == 0):
                streams[k].wait_event(ev_stage[(k - 1) % P])
            streams[k].wait_event(ev_match[s])  # the matcher that read this set (step i-2) and copied its last frame is done
            exts[k].extract_batch_device(frames[(i % 2) * B + k * n:].data_ptr(), n, W, H, W, W * H, kps[s].data_ptr() + KP * (1 + k * n),
                                         desc[s].data_ptr() + DS * (1 + k * n), cap, nout[s].data_ptr() + 4 * (1 + k * n), streams[k].cuda_stream)
            ev_done[s][k].record(streams[k])
        if with_match:
            with torch.cuda.stream(s_match):
                for k in range(P):
                    s_match.wait_event(ev_done[s][k])
                matcher.match(B, cap, desc[s].data_ptr(), kps[s].data_ptr() + 12, None, nout[s].data_ptr(), desc[s].data_ptr() + DS,
                              kps[s].data_ptr() + KP + 12, nout[s].data_ptr() + 4, 28, 50, 0.7, True, mb.data_ptr(), nm.data_ptr(), s_match.cuda_stream)
                kps[1 - s][0].copy_(kps[s][B], non_blocking=True)
                desc[1 - s][0].copy_(desc[s][B], non_blocking=True)
                nout[1 - s][0:1].copy_(nout[s][B:B + 1], non_blocking=True)
                ev_match[s].record(s_match)
    return step, lambda: (int(nout[0][1:].sum()), int(nm.sum()))


configs = [(1, "-"), (2, "pyramid"), (2, "C"), (4, "pyramid"), (4, "C"), (2, "fused-pyramid"), (4, "fused-pyramid")]
for rep in range(2):
    for P, after in configs:
        step, total = run(P, after)
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print("%d parts %-16s %.3f ms per 512 frames  %.0f frames/s  %s" % (P, after, dt * 1e3, B / dt, total()), flush=True)
        del step, total
        torch.cuda.empty_cache()
