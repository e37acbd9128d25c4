"""Multi-GPU sampling = independent pockets sharded across ranks, no data-path collective
(every pocket is an independent graph; the reference's outer loop is sample.py:159).
One process per GPU; torch.distributed (backend 'nccl' = RCCL on ROCm, 'gloo' in CPU tests) is used
only for the barrier and for reducing timing / throughput counters."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_indices(n_items, rank, world):
    """Pocket indices of this rank: i = rank (mod world) -- round-robin so ragged pocket sizes average out."""
    return list(range(rank, n_items, world))


def init_process_group(backend=None):
    rank, world, local = env_rank_world()
    # CBGX_DIST_FORCE=1: create the group even for one rank, so that a 1-GPU box runs the barrier and the reductions through
    # RCCL itself (tests/test_gpu_bench.py) -- the code path of the N-GPU job minus the peers
    force = os.environ.get("CBGX_DIST_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("CBGX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # rendezvous, in this order: (1) the FileStore path of cbgbench_amd.launch (no port exists to collide on); (2) an outer
        # launcher's MASTER_ADDR / MASTER_PORT (torchrun, the driver's N-GPU entry); (3) a one-rank group has nobody to meet: a
        # private FileStore.  There is no default port: N > 1 ranks without either are a launch error, said so.
        rdzv = os.environ.get("CBGX_RDZV_FILE")
        if rdzv is None and "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError(f"{world} ranks but no rendezvous: start them with `python -m cbgbench_amd.launch --nproc {world} ...` "
                                   f"(file rendezvous, no port) or under torch.distributed.run (MASTER_ADDR / MASTER_PORT)")
            import tempfile
            rdzv = os.path.join(tempfile.mkdtemp(prefix="cbgx_rdzv_"), "store")
        if rdzv is not None:
            dist.init_process_group(backend=backend, init_method="file://" + rdzv, rank=rank, world_size=world)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":      # name the device: the communicator is created lazily at the first collective
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def reduce_max_sum(elapsed_s, units, device="cpu"):
    """(max over ranks of elapsed, sum over ranks of units): whole-job throughput = units_sum / elapsed_max."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), float(units)
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_values(value, device="cpu"):
    """[value of rank 0, value of rank 1, ...] on every rank (one all_gather of a double): the per-rank throughputs behind a
    max / sum pair, so that a straggling rank is visible in the N > 1 bench line."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    if dist.get_backend() == "gloo":
        device = "cpu"
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
