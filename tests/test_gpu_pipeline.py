"""orbgpu_pipeline (staggered sub-batches on streams of their own) against the plain batched call: bit-identical
key points, descriptors and counts; events order the consumer.  include/orbgpu.h, `orbgpu_pipeline_*`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(n, w=320, h=240, seed=5):
    from orb_slam2_map_amd.synth import Stream
    st = Stream(w, h, seed)
    return np.stack([st.frame(t)[0] for t in range(n)])


@pytest.mark.parametrize("parts,batch", [(2, 8), (3, 8), (4, 13), (4, 3), (1, 5)])
def test_pipeline_matches_plain_batch(parts, batch):
    import torch
    from orb_slam2_map_amd import lib as G
    w, h = 320, 240
    host = _frames(batch, w, h)
    frames = torch.from_numpy(host).cuda()
    ref = G.ORBextractor(500, max_batch=batch)
    cap = ref.max_keypoints(w, h)
    out = [[torch.zeros((batch, cap, 7), dtype=torch.float32, device="cuda"), torch.zeros((batch, cap, 32), dtype=torch.uint8, device="cuda"),
            torch.zeros(batch, dtype=torch.int32, device="cuda")] for _ in range(2)]
    s = torch.cuda.current_stream()
    ref.extract_batch_device(frames.data_ptr(), batch, w, h, w, w * h, out[0][0].data_ptr(), out[0][1].data_ptr(), cap,
                             out[0][2].data_ptr(), s.cuda_stream)
    pl = G.ExtractorPipeline(500, max_batch=batch, parts=parts)
    assert pl.max_keypoints(w, h) == cap and len(pl.parts) == parts
    done = torch.cuda.Event()
    done.record(s)  # torch creates the handle on first use
    for rep in range(3):  # later calls chain on the previous call's last part
        pl.extract_batch_device(frames.data_ptr(), batch, w, h, w, w * h, out[1][0].data_ptr(), out[1][1].data_ptr(), cap,
                                out[1][2].data_ptr(), None, done.cuda_event)
    s.wait_event(done)
    torch.cuda.synchronize()
    n0, n1 = out[0][2].cpu().numpy(), out[1][2].cpu().numpy()
    assert (n0 > 0).all() and np.array_equal(n0, n1)
    for f in range(batch):
        assert torch.equal(out[0][0][f, :n0[f]].view(torch.int32), out[1][0][f, :n0[f]].view(torch.int32))  # bit patterns (class_id -1 is a NaN as float)
        assert torch.equal(out[0][1][f, :n0[f]], out[1][1][f, :n0[f]])
    # stream-ordered use: wait() makes the consumer stream see the results
    out[1][2].zero_()
    pl.extract_batch_device(frames.data_ptr(), batch, w, h, w, w * h, out[1][0].data_ptr(), out[1][1].data_ptr(), cap, out[1][2].data_ptr())
    pl.wait(s.cuda_stream)
    got = out[1][2].clone()  # on s, after the wait
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), n0)
    # the parts are ordinary handles: per-stage profiling works on them
    pl.parts[0].set_profiling(True)
    pl.extract_batch_device(frames.data_ptr(), batch, w, h, w, w * h, out[1][0].data_ptr(), out[1][1].data_ptr(), cap, out[1][2].data_ptr())
    pl.wait(s.cuda_stream)
    torch.cuda.synchronize()
    t = pl.parts[0].stage_times()
    assert set(t) == {"pyramid", "fast", "quadtree", "orient", "blur", "describe"} and all(v > 0 for v in t.values())
    pl.close()


def test_pipeline_wait_event_and_errors():
    import torch
    from orb_slam2_map_amd import lib as G
    w, h, batch = 320, 240, 6
    frames = torch.from_numpy(_frames(batch, w, h, 9)).cuda()
    pl = G.ExtractorPipeline(500, max_batch=batch, parts=2)
    cap = pl.max_keypoints(w, h)
    kps = torch.zeros((batch, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((batch, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(batch, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    gate = torch.cuda.Event()
    staged = torch.empty_like(frames)
    with torch.cuda.stream(side):
        staged.copy_(frames, non_blocking=True)  # the "upload" the parts must wait for
        gate.record(side)
    pl.extract_batch_device(staged.data_ptr(), batch, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), gate.cuda_event)
    pl.wait(0)
    torch.cuda.synchronize()
    ref = G.ORBextractor(500, max_batch=batch)
    k2, d2, n2 = torch.zeros_like(kps), torch.zeros_like(desc), torch.zeros_like(nout)
    ref.extract_batch_device(frames.data_ptr(), batch, w, h, w, w * h, k2.data_ptr(), d2.data_ptr(), cap, n2.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(nout, n2)
    with pytest.raises(G.OrbGpuError):
        pl.extract_batch_device(0, batch, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr())
    with pytest.raises(G.OrbGpuError):
        G.ExtractorPipeline(500, max_batch=4, parts=0)


def test_concurrent_blur_gives_the_same_frames(gpu, stream640):
    """orbgpu_extractor_set_concurrent_blur: the blur on a stream of the handle's own, forked after the pyramid and joined in
    front of the descriptor stage -- identical key points and descriptors, with and without profiling events, and after
    switching back."""
    torch = pytest.importorskip("torch")
    B, w, h = 8, 640, 480
    imgs = torch.from_numpy(stream640.gray_batch(20, B)).cuda()
    ge = gpu.ORBextractor(1000, max_batch=B)
    cap = ge.max_keypoints(w, h)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        nout = torch.zeros(B, dtype=torch.int32, device="cuda")
        ge.extract_batch_device(imgs.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
        torch.cuda.synchronize()
        return kps.cpu().numpy().view(np.int32), desc.cpu().numpy(), nout.cpu().numpy()  # (class_id = -1 is a NaN as float)
    ref = run()
    assert ref[2].min() > 500
    ge.set_concurrent_blur(True)
    for prof in (False, True, False):
        ge.set_profiling(prof)
        for _ in range(3):
            got = run()
            assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    ge.set_profiling(True)
    run()
    t = ge.stage_times()
    assert t["blur"] > 0 and t["fast"] > 0
    ge.set_profiling(False)
    ge.set_concurrent_blur(False)
    got = run()
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
