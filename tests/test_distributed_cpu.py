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


def _syncbn_worker(rank, world, port, q):
    """The exchange step of SyncBatchNorm as the training forward runs it (train_net._sync_world, train_ops._all_reduce_sums):
    per-rank {sum, sum of squares} doubles -> all-reduce -> the full-batch mean / biased variance on every rank."""
    import numpy as np, torch
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tn = importlib.import_module(PKG + ".keypoints.train_net")
    ops = importlib.import_module(PKG + ".keypoints.train_ops")
    tn._SYNC[0] = True
    before = tn._sync_world()          # no process group yet: plain BatchNorm, like torch's SyncBatchNorm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    with_group = tn._sync_world()
    tn._SYNC[0] = None
    unset = tn._sync_world()
    x = np.random.default_rng(5).normal(0.3, 2.0, (4, 6, 5, 8))  # [B, C, H, W], the same on both ranks
    mine = x[rank * 2:(rank + 1) * 2]
    sums = torch.from_numpy(np.stack([mine.sum((0, 2, 3)), (mine ** 2).sum((0, 2, 3))], 1).reshape(-1).copy())
    ops._all_reduce_sums(sums, None)
    count = mine[:, 0].size * world
    s = sums.numpy().reshape(-1, 2)
    mean, var = s[:, 0] / count, s[:, 1] / count - (s[:, 0] / count) ** 2
    q.put((rank, before is None, with_group == (None, world), unset is None,
           bool(np.allclose(mean, x.mean((0, 2, 3)), rtol=1e-12, atol=1e-12) and np.allclose(var, x.var((0, 2, 3)), rtol=1e-10, atol=1e-12))))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_exchange_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 2000
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == [(0, True, True, True, True), (1, True, True, True, True)]


def test_shard_range_partitions():
    d = importlib.import_module(PKG + ".keypoints.distributed")
    for n in (0, 1, 5, 8, 5000):
        for w in (1, 2, 3, 8):
            parts = [list(d.shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_bench_multi_rank_launch_contract_two_ranks():
    """bench.py under the driver's N > 1 command line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`),
    rehearsed on CPU over gloo (HH_BENCH_REHEARSAL=cpu swaps the GPU step for a rank-dependent sleep; process-group init,
    the barrier/synchronize bracket, the MAX over ranks and the rank-0 line are the code the real multi-GPU run executes)."""
    import json
    import subprocess
    port = 29500 + (os.getpid() + 191) % 2000
    env = dict(os.environ, HH_BENCH_REHEARSAL="cpu", OMP_NUM_THREADS="1")
    steps, warmup, batch = 5, 1, 32
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one line, from rank 0
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in r, key
    assert r["n_gpus"] == 2 and r["steps"] == steps and r["warmup"] == warmup and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["global_batch"] == 2 * batch and r["vs_baseline"] is None
    # rank 1 sleeps 4 ms per step, rank 0 2 ms: the reported time is the slower rank's (MAX over ranks) ...
    assert r["ms_per_step"] >= 4.0
    # ... and the value is the whole-job aggregate over both ranks for that time
    assert abs(r["value"] - 2 * batch / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-3


def test_bench_gpus_flag_without_a_launcher_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no torch.distributed.run around it): bench.py starts the two ranks itself before touching
    a GPU and prints the N = 2 line; with a WORLD_SIZE that contradicts --gpus it refuses instead of reporting the wrong n_gpus."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HH_BENCH_REHEARSAL="cpu", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["config"]["global_batch"] == 64
    # a launcher that gives one rank to a command line asking for eight: no line, non-zero exit
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 8 but WORLD_SIZE=1" in bad.stderr
    # more GPUs than the node has (none here): refused before anything is launched
    env2 = {k: v for k, v in env.items() if k != "HH_BENCH_REHEARSAL"}
    none = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                          env=env2, capture_output=True, text=True, timeout=120)
    assert none.returncode != 0 and "shows" in none.stderr and not [l for l in none.stdout.splitlines() if l.startswith("{")]
