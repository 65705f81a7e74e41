"""World-size-2 (and 3) gloo tests of the sample sharding + token-stream gather used at N > 1 GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream(i):
    rng = np.random.default_rng(100 + i)
    return rng.integers(3, 518, size=5 + 7 * (i % 4)).astype(np.int64)


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from edgerunner_amd import dist as D
    r, w, _ = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    mine = D.shard_indices(n_items, rank, world)
    full = D.gather_token_streams([_stream(i) for i in mine], n_items)
    ok = len(full) == n_items and all(np.array_equal(full[i], _stream(i)) for i in range(n_items))
    t = D.max_over_ranks(float(rank + 1))
    by_rank = D.all_ranks(10.0 * rank + 0.5)              # bench.py's per-rank diagnostics: every rank sees every rank's value, in rank order
    ok = ok and by_rank == [10.0 * r_ + 0.5 for r_ in range(world)]
    D.barrier()
    q.put((rank, bool(ok), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (2, 2), (3, 7), (2, 1)])
def test_gather_token_streams_gloo(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(t == float(world) for _, _, t in res)


def test_shard_indices_partition():
    from edgerunner_amd.dist import shard_indices
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 256):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_plan_and_group_jobs_cover_the_reference_loops():
    """infer.py's sharding of the reference's serial loops (infer.py:99-101,136-137): every (path, repeat, num_face) job runs on
    exactly one rank, one rank per path exports the cloud, and a batched call holds one face count and one cloud size."""
    from edgerunner_amd import dist as D
    paths = [f"in/{c}.npy" for c in "abcde"]
    npts = {p: (300 if p.endswith("c.npy") else 512) for p in paths}
    faces = (1000, 4000)
    for world in (1, 2, 3, 8):
        seen, owners = [], {}
        for rank in range(world):
            jobs, mine, pc_owner = D.plan_jobs(paths, 3, faces, rank, world)
            assert jobs[:4] == [(paths[0], 0, 1000), (paths[0], 0, 4000), (paths[0], 1, 1000), (paths[0], 1, 4000)]
            seen += mine
            owners.update(pc_owner)
            covered = []
            for nf, chunk in D.group_jobs(jobs, mine, lambda p: npts[p], 4):
                assert 1 <= len(chunk) <= 4
                assert {jobs[j][2] for j in chunk} == {nf} and len({npts[jobs[j][0]] for j in chunk}) == 1
                covered += chunk
            assert sorted(covered) == sorted(mine)
        assert sorted(seen) == list(range(len(paths) * 3 * len(faces)))
        assert set(owners) == set(paths) and all(0 <= r < world for r in owners.values())
        for k, p in enumerate(paths):          # the owner is the rank that runs the path's first job
            assert owners[p] == (k * 3 * len(faces)) % world
