"""Gallery sharding and the one exchange step (all_gather of per-shard top-k + merge), without a GPU: pure host logic plus a
world_size-2 gloo run of the same gather the GPU bench does over RCCL."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

SH = importlib.import_module("msu-latentafis_amd.host.sharding")


def brute_topk(scores, k):
    Q, G = scores.shape
    oi = np.zeros((Q, k), np.int64); os_ = np.zeros((Q, k), np.float32)
    for q in range(Q):
        order = np.lexsort((np.arange(G), -scores[q].astype(np.float64)))[:k]
        oi[q] = order; os_[q] = scores[q][order]
    return oi, os_


def shard_topk(scores, lo, hi, k):
    i, s = brute_topk(scores[:, lo:hi], min(k, hi - lo))
    pi = np.full((scores.shape[0], k), -1, np.int64); ps = np.full((scores.shape[0], k), -np.inf, np.float32)
    pi[:, :i.shape[1]] = i + lo; ps[:, :s.shape[1]] = s
    return pi, ps


def test_shard_bounds_tile_and_balance():
    rng = np.random.default_rng(0)
    cost = rng.integers(600, 1001, 10007)
    for world in (1, 2, 3, 8):
        b = SH.shard_bounds(cost, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == len(cost)
        assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
        loads = [cost[lo:hi].sum() for lo, hi in b]
        assert max(loads) - min(loads) <= 2 * cost.max()
    assert SH.shard_bounds(np.ones(3), 8)[-1][1] == 3          # more ranks than templates: empty shards are legal


@pytest.mark.parametrize("world", [2, 8])
def test_merge_equals_global_topk(world):
    rng = np.random.default_rng(1)
    Q, G, k = 5, 1000, 24
    scores = np.round(rng.random((Q, G)).astype(np.float32) * 50, 0)      # many ties: the index tie-break must hold across shards
    scores[:, ::7] = -1.0
    b = SH.shard_bounds(np.ones(G), world)
    per = [shard_topk(scores, lo, hi, k) for lo, hi in b]
    mi, ms = SH.merge_topk(np.stack([p[0] for p in per]), np.stack([p[1] for p in per]), k)
    wi, ws = brute_topk(scores, k)
    assert np.array_equal(mi, wi) and np.array_equal(ms, ws)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    Q, G, k = 4, 400, 24
    scores = np.round(rng.random((Q, G)).astype(np.float32) * 30, 0)
    lo, hi = SH.shard_bounds(np.ones(G), world)[rank]
    pi, ps = shard_topk(scores, lo, hi, k)
    mi, ms = SH.gather_topk(pi, ps, k)
    wi, ws = brute_topk(scores, k)
    q.put((rank, bool(np.array_equal(mi, wi) and np.array_equal(ms, ws))))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_topk_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
