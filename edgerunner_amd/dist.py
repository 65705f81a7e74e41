"""Multi-GPU sharding of independent samples.

The reference generates strictly serially (``infer.py:99-101,136-137`` loop files x
``test_repeat`` x ``test_num_face``; ``LMM.generate`` asserts B == 1) and has no
inference-time parallelism.  Samples are independent, so this path shards them
block-cyclically over one process per GPU - each rank holds a full weight replica
and its own KV cache, no data-path collective - and performs ONE exchange at the
end: an all-gather of the padded token streams (RCCL over xGMI when the backend is
``nccl``; ``gloo`` in the CPU tests).  The message is tiny (<= a few MB), i.e.
latency-bound, so it is a single fused tensor: ids and length travel together.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def _collectives_on() -> bool:
    """Collectives run when there is more than one rank - or, for the single-GPU smoke test of the RCCL path
    (tests/test_gpu_scripts.py), when ER_DIST_FORCE=1 keeps them on in a one-rank group."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("ER_DIST_FORCE") == "1")


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Sample i runs on rank i mod world (block-cyclic keeps per-rank work balanced)."""
    return list(range(rank, n_items, world))


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """One process per GPU, rendezvous from the torchrun env (MASTER_ADDR must be 127.0.0.1 on this pool)."""
    rank, world, local = env_rank_world()
    if (world > 1 or os.environ.get("ER_DIST_FORCE") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def pack_streams(streams: Sequence[np.ndarray], rows: int, width: int, pad: int = 0) -> torch.Tensor:
    """[rows, width+1] int32: token ids padded with `pad`, last column = true length (-1 = empty row)."""
    out = torch.full((rows, width + 1), pad, dtype=torch.int32)
    out[:, width] = -1
    for j, s in enumerate(streams):
        s = np.asarray(s)
        out[j, : len(s)] = torch.from_numpy(s.astype(np.int32))
        out[j, width] = len(s)
    return out


def gather_token_streams(local_streams: Sequence[np.ndarray], n_items: int, device=None, pad: int = 0) -> List[np.ndarray]:
    """All ranks contribute the streams of their ``shard_indices``; every rank gets the
    full list back in global sample order.  One all-reduce (max length) + one all-gather."""
    if not _collectives_on():
        return [np.asarray(s) for s in local_streams]
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_indices(n_items, rank, world)
    assert len(mine) == len(local_streams), (len(mine), len(local_streams))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rows = (n_items + world - 1) // world
    width = torch.tensor([max([len(s) for s in local_streams], default=0)], dtype=torch.int32, device=device)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    width = int(width.item())
    payload = pack_streams(local_streams, rows, width, pad).to(device)
    gathered = torch.empty((world * rows, width + 1), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(gathered, payload)
    g = gathered.view(world, rows, width + 1).cpu().numpy()
    out: List[Optional[np.ndarray]] = [None] * n_items
    for r in range(world):
        for j, i in enumerate(shard_indices(n_items, r, world)):
            n = int(g[r, j, width])
            out[i] = g[r, j, :n].astype(np.int64)
    return out  # type: ignore[return-value]


def barrier():
    if _collectives_on():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not _collectives_on():
        return value
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, device=None) -> List[float]:
    """The same scalar from every rank, in rank order (one all-gather of 8 bytes per rank; diagnostics of bench.py --gpus N)."""
    if not _collectives_on():
        return [float(value)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


# ------------------------------------------------------------------------------------------------
# Job planning of the drop-in scripts (reference infer.py:99-101,136-137: serial loops path x repeat x num_face)
def plan_jobs(paths: Sequence[str], test_repeat: int, test_num_face: Sequence[int], rank: int, world: int):
    """-> (jobs, mine, pc_owner): every (path, repeat index, num_faces) job in the reference's loop order, the indices
    this rank runs (block-cyclic), and for every path the ONE rank that exports its point cloud (the rank owning the
    path's first job)."""
    jobs = [(p, i, nf) for p in paths for i in range(test_repeat) for nf in test_num_face]
    per_path = test_repeat * len(test_num_face)
    pc_owner = {p: (k * per_path) % world for k, p in enumerate(paths)}
    return jobs, shard_indices(len(jobs), rank, world), pc_owner


def group_jobs(jobs, mine: Sequence[int], n_points_of, rows_max: int):
    """Chunks of this rank's jobs that can share ONE batched generate() call: same face count (one bucket embedding per
    call) and same number of points (one dense [B, N, 3] tensor), at most rows_max rows.  -> [(num_faces, [job index])]."""
    groups = {}
    for j in mine:
        groups.setdefault((jobs[j][2], n_points_of(jobs[j][0])), []).append(j)
    out = []
    for (nf, _), members in groups.items():
        for c0 in range(0, len(members), max(1, rows_max)):
            out.append((nf, members[c0:c0 + max(1, rows_max)]))
    return out
