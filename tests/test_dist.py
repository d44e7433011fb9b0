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


def _run_bench(extra, timeout=900):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, cwd=root,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_self_spawns_ranks_without_a_gpu():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) must start 2 ranks as a child process group.
    Without a GPU every rank refuses loudly (no CPU fallback) and the parent relays the failure."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_two_ranks_on_one_gpu")
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--rehearse-on-device0", "--steps", "1", "--warmup", "0",
                    "--batch", "2", "--pool", "2", "--no-cpu-baseline"], timeout=300)
    assert r.returncode != 0
    assert "no GPU visible" in r.stderr or "no HIP device" in r.stderr, r.stderr[-2000:]
    assert "launch with torch.distributed.run" not in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The N>1 code path of bench.py on real hardware: the driver's own command form (`bench.py --gpus 2`),
    two ranks rehearsed on device 0 with gloo carrying the barrier / reductions."""
    import json
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--rehearse-on-device0", "--steps", "2", "--warmup", "1",
                    "--batch", "32", "--pool", "64", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["sequences"] == 2
    assert out["steps"] == 2 and out["scaling"] == "weak"
    frames = out["value"] * out["ms_per_step"] * 1e-3 * out["steps"]
    assert abs(frames - 2 * 2 * 32) < 1e-6 * frames + 1e-3  # frames summed over both ranks
    assert "cpu_baseline" not in out


@pytest.mark.gpu
def test_bench_c3_batch_two_ranks_on_one_gpu():
    """north_star's 1280x960 line under --gpus N (`--workload c3_batch`, Tracking.cc:1447-1497 per frame): two ranks rehearsed
    on device 0, 4 sequences each; the contract fields, the workload name and the frame count over both ranks."""
    import json
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--rehearse-on-device0", "--workload", "c3_batch", "--steps", "2",
                    "--warmup", "1", "--batch", "4", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["sequences"] == 8 and out["config"]["workload"].startswith("C3")
    assert "1280x960" in out["metric"] and out["scaling"] == "weak" and out["steps"] == 2
    frames = out["value"] * out["ms_per_step"] * 1e-3 * out["steps"]
    assert abs(frames - 2 * 2 * 4) < 1e-6 * frames + 1e-3
    assert out["matches_per_frame"] > 500 and out["keypoints_per_frame"] > 1900
    assert len(out["timed_region_repeats"]["frames_per_s"]) == 3 and out["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_c3_batch_single_rank_with_cpu_baseline():
    import json
    r = _run_bench(["--workload", "c3_batch", "--steps", "2", "--warmup", "1", "--batch", "8"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["sequences"] == 8
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 1


def _single_rank(backend, device, port):
    """Body of the one-rank tests; runs in a child process so that the process group never leaks into pytest."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = '%d'\n"
        "import torch\n"
        "import torch.distributed as dist\n"
        "from orb_slam2_map_amd import dist as D\n"
        "if %r == 'cuda': torch.cuda.set_device(0)\n"
        "D.init(%r, 0, 1, force_group=True)\n"
        "assert dist.is_initialized() and dist.get_backend() == %r and dist.get_world_size() == 1\n"
        "D.barrier(1)\n"
        "t, n = D.aggregate(1.25, 640, 1, device=%r)\n"
        "assert (t, n) == (1.25, 640.0), (t, n)\n"
        "D.barrier(1)\n"
        "D.finalize(1)\n"
        "assert not dist.is_initialized()\n"
        "print('single-rank %s ok')\n" % (root, port, device, backend, backend, device, backend))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0 and "single-rank %s ok" % backend in r.stdout, r.stdout[-3000:]


def test_forced_single_rank_group_gloo():
    """force_group: init / barrier / all-reduce / finalize go through the backend for one rank as well."""
    _single_rank("gloo", "cpu", 29900 + os.getpid() % 90)


@pytest.mark.gpu
def test_rccl_single_rank_on_gpu():
    """The RCCL branches of dist.py (communicator bound to the rank's device, barrier with device_ids, all-reduce of
    CUDA tensors) executed on hardware: a one-rank communicator is what a one-GPU box can run of them."""
    _single_rank("nccl", "cuda", 29800 + os.getpid() % 90)


@pytest.mark.gpu
def test_bench_single_rank_through_rccl():
    """bench.py itself with its barrier / reductions going through RCCL (world 1, --force-group)."""
    import json
    r = _run_bench(["--backend", "nccl", "--force-group", "--steps", "2", "--warmup", "1", "--batch", "32", "--pool", "64",
                    "--no-cpu-baseline", "--no-secondary"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
