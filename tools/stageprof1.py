"""Per-stage extraction times of single frames (B = 1): python tools/stageprof1.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream

for (W, H, NF) in ((640, 480, 1000), (1280, 960, 2000)):
    st = Stream(W, H, 1234)
    frames = torch.from_numpy(np.stack([st.frame(t)[0] for t in range(8)])).cuda()
    ext = G.ORBextractor(NF, max_batch=1)
    cap = ext.max_keypoints(W, H)
    kps = torch.zeros((1, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for i in range(5):
        ext.extract_batch_device(frames[i % 8:].data_ptr(), 1, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50):
        ext.extract_batch_device(frames[i % 8:].data_ptr(), 1, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / 50 * 1e3
    ext.set_profiling(True)
    for i in range(20):
        ext.extract_batch_device(frames[i % 8:].data_ptr(), 1, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
    torch.cuda.synchronize()
    t = ext.stage_times()
    print(W, H, json.dumps({k: round(v * 1e3, 1) for k, v in t.items()}), "sum_us", round(sum(t.values()) * 1e3, 1), "back_to_back_us", round(total, 1))
