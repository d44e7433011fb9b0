"""Multi-GPU harness: one process per GPU, one independent RGB-D sequence per rank.

The hot path has no exchange step -- extraction is per frame and matching is per sequence -- so
there is NO data-path collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests) carries only the start barrier and the two tiny reductions that form
the aggregate throughput: max elapsed time over ranks and the sum of processed frames.
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def sequence_seed(base_seed, rank):
    """Seed of the synthetic sequence rank `rank` owns (frames shard by sequence, never by frame)."""
    return base_seed + 1000 * rank


def shard_sequences(n_sequences, rank, world):
    """Sequences owned by `rank` when there are more sequences than ranks (round-robin)."""
    return [s for s in range(n_sequences) if s % world == rank]


def _group_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def init(backend, rank, world, force_group=False):
    """Process group for world > 1.  `force_group` builds it for a single rank as well: the barrier / reductions then run
    through the backend exactly as they do on the 8-GPU node (a one-rank RCCL communicator is what a one-GPU box can
    execute -- tests/test_dist.py::test_rccl_single_rank_on_gpu)."""
    import torch.distributed as dist
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":  # RCCL: bind the communicator to this rank's device up front
            import torch
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)


def barrier(world):
    import torch.distributed as dist
    if _group_up():
        if dist.get_backend() == "nccl":  # RCCL: name the rank's own device instead of letting the backend guess one
            import torch
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def aggregate(elapsed_s, frames, world, device="cpu"):
    """(max elapsed over ranks, total frames over ranks)."""
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    n = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    if _group_up():
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())


def finalize(world):
    import torch.distributed as dist
    if _group_up():
        dist.destroy_process_group()
