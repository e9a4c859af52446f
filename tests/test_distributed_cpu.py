"""World-size-2 gloo test of the image sharding used for multi-GPU evaluation (no collective on the data path)."""
import importlib
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module(PKG + ".keypoints.distributed")
    mine = d.shard_range(n_items, rank, world)
    local = [{"image_id": i, "score": i * 0.5} for i in mine]  # what eval.py:37-49 packs per image
    allr = d.gather_results(local)
    if rank == 0:
        q.put([r["image_id"] for r in allr])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [0, 1, 7, 10])
def test_shard_and_gather_two_ranks(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_items) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == list(range(n_items))


def test_shard_range_partitions():
    d = importlib.import_module(PKG + ".keypoints.distributed")
    for n in (0, 1, 5, 8, 5000):
        for w in (1, 2, 3, 8):
            parts = [list(d.shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
