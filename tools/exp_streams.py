#!/usr/bin/env python3
"""Experiment: extraction throughput with the batch split over S concurrent streams/handles."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream

W, H = 640, 480
B, POOL = 256, 1024
st = Stream(W, H, 1234)
frames = torch.from_numpy(np.stack([st.frame(t)[0] for t in range(POOL)])).cuda()
for S in (1, 2, 4):
    b = B // S
    exts = [G.ORBextractor(1000, max_batch=b) for _ in range(S)]
    cap = exts[0].max_keypoints(W, H)
    streams = [torch.cuda.Stream() for _ in range(S)]
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")

    def step(i):
        base = (i * B) % POOL
        for s in range(S):
            src = frames[base + s * b: base + (s + 1) * b]
            exts[s].extract_batch_device(src.data_ptr(), b, W, H, W, W * H, kps[s * b].data_ptr(), desc[s * b].data_ptr(),
                                         cap, nout[s * b:].data_ptr(), streams[s].cuda_stream)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for i in range(K):
        step(3 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams %d: %.3f ms/step of %d frames -> %.0f fps (extract only)" % (S, dt / K * 1e3, B, B * K / dt), flush=True)
    del exts
