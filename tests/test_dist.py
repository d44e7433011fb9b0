"""N>1 path on CPU: world_size-2 gloo run of the multi-GPU harness (orb_slam2_map_amd/dist.py)."""
import os
import zlib

import numpy as np
import pytest


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from orb_slam2_map_amd import dist as D
    from orb_slam2_map_amd.synth import Stream
    D.init("gloo", rank, world)
    st = Stream(640, 480, D.sequence_seed(1234, rank))
    crc = zlib.crc32(st.frame(0)[0].tobytes())
    D.barrier(world)
    elapsed, frames = D.aggregate(0.5 + rank, 100 * (rank + 1), world)
    q.put((rank, crc, elapsed, frames, D.shard_sequences(5, rank, world)))
    D.finalize(world)


def test_two_rank_gloo_harness():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, crc0, t0, n0, s0), (r1, crc1, t1, n1, s1) = res
    assert crc0 != crc1, "each rank owns its own sequence"
    assert t0 == t1 == 1.5 and n0 == n1 == 300.0  # max over ranks, sum over ranks
    assert s0 == [0, 2, 4] and s1 == [1, 3]


def test_single_rank_is_a_noop():
    from orb_slam2_map_amd import dist as D
    assert D.aggregate(2.0, 7, 1) == (2.0, 7.0)
    assert D.sequence_seed(1234, 3) == 4234
