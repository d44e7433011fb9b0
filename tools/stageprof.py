"""Per-stage extraction times at B = 256 (serial schedule), for quick kernel iterations: python tools/stageprof.py
With --single: single frames (B = 1) at 640x480 / 1000 and 1280x960 / 2000 features, stage times and back-to-back time."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream



def single():
    for (W, H, NF) in ((640, 480, 1000), (1280, 960, 2000)):
        st = Stream(W, H, 1234)
        frames = torch.from_numpy(np.stack([st.frame(t)[0] for t in range(8)])).cuda()
        ext = G.ORBextractor(NF, max_batch=1)
        cap = ext.max_keypoints(W, H)
        kps = torch.zeros((1, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
        nout = torch.zeros(1, dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream

        def run(n):
            for i in range(n):
                ext.extract_batch_device(frames[i % 8:].data_ptr(), 1, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap,
                                         nout.data_ptr(), s)
        run(5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(50)
        e1.record()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1) / 50 * 1e3
        ext.set_profiling(True)
        run(20)
        torch.cuda.synchronize()
        t = ext.stage_times()
        print(W, H, json.dumps({k: round(v * 1e3, 1) for k, v in t.items()}), "sum_us", round(sum(t.values()) * 1e3, 1),
              "back_to_back_us", round(total, 1))


if "--single" in sys.argv:
    single()
    sys.exit(0)
W, H, B, POOL = 640, 480, 256, 1024
st = Stream(W, H, 1234)
frames = torch.from_numpy(np.stack([st.frame(t)[0] for t in range(POOL)])).cuda()
ext = G.ORBextractor(1000, max_batch=B)
cap = ext.max_keypoints(W, H)
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
nout = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for i in range(3):
    ext.extract_batch_device(frames[(i * B) % POOL:].data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
torch.cuda.synchronize()
ext.set_profiling(True)
for i in range(20):
    ext.extract_batch_device(frames[(i * B) % POOL:].data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
torch.cuda.synchronize()
t = ext.stage_times()
print(json.dumps({k: round(v * 1e3, 1) for k, v in t.items()}), "sum_us", round(sum(t.values()) * 1e3, 1),
      "crc", int(kps.view(torch.int32).sum().item()) & 0xFFFFFFFF, int(desc.sum().item()))
