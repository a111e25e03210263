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


# ---- the C++ host's multi-rank pieces (msu-latentafis_amd/csrc/rank_exchange.cpp), without a GPU ---------------------------------
def _match_exe():
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "msu-latentafis_amd", "csrc")
    exe = os.path.join(csrc, "match_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", csrc, "match_selftest"], check=True)
    return exe


def test_cpp_shard_bounds_equal_python(tmp_path):
    import subprocess
    exe = _match_exe()
    rng = np.random.default_rng(9)
    for G, world in ((1000, 8), (13, 4), (5, 8), (1, 2)):
        w = rng.integers(0, 1000, G).astype(np.int32)
        w[rng.integers(0, G, max(1, G // 10))] = 0                      # templates without texture
        f = tmp_path / f"w_{G}_{world}.txt"; f.write_text("\n".join(str(int(x)) for x in w) + "\n")
        out = subprocess.run([exe, "-selftest-shards", str(f), "-world", str(world)], capture_output=True, text=True, check=True).stdout.split()
        got = [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(2 * world)]
        assert got[:world] == SH.shard_bounds(w, world), (G, world)
        assert got[world:] == SH.shard_bounds(np.ones(G), world), (G, world)


def test_cpp_rendezvous_three_ranks():
    """The ncclUniqueId hand-off of `match` (rank 0 -> every rank over TCP) with three local processes, rank 0 started last."""
    import subprocess, socket, time
    exe = _match_exe()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = lambda r: dict(os.environ, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port - 1))
    procs = [subprocess.Popen([exe, "-selftest-exchange"], env=env(r), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (1, 2)]
    time.sleep(0.5)                                                     # the peers retry until rank 0 listens
    procs.append(subprocess.Popen([exe, "-selftest-exchange"], env=env(0), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=60) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    ids = {o[0].strip().split(" id ")[1] for o in outs}
    assert len(ids) == 1 and sorted(o[0].split()[1] for o in outs) == ["0", "1", "2"]


def test_cpp_tcp_all_gather_and_agreement_three_ranks():
    """AFIS_EXCHANGE=tcp: the gather-and-return through rank 0 that lets `match` run N > 1 ranks on one GPU, and the agreement point
    that stops every rank when one reports a failure (three local processes, three exchanges + one agreement each)."""
    import subprocess, socket
    exe = _match_exe()
    for fail_rank, want in ((0, "agree 0"), (2, "agree 7")):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        env = lambda r: dict(os.environ, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port - 1),
                             AFIS_EXCHANGE_TIMEOUT_S="30")
        procs = [subprocess.Popen([exe, "-selftest-allgather", "-fail-rank", str(fail_rank)], env=env(r), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for r in (2, 1, 0)]
        outs = [p.communicate(timeout=90) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        assert all(want in o[0] and "gathered ok" in o[0] for o in outs), outs


def _cpp_exchange_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      AFIS_EXCHANGE="tcp", AFIS_EXCHANGE_TIMEOUT_S="30")
    x = SH.CppExchange(0)                                               # tcp stand-in: no device is touched
    rng = np.random.default_rng(3)
    Q, G, k = 4, 400, 24
    scores = np.round(rng.random((Q, G)).astype(np.float32) * 30, 0)
    lo, hi = SH.shard_bounds(np.ones(G), world)[rank]
    pi, ps = shard_topk(scores, lo, hi, k)
    ok = True
    for _ in range(3):                                                  # the exchange counter must stay in step over repeated steps
        mi, ms = x.gather_topk(pi, ps, k)
        wi, ws = brute_topk(scores, k)
        ok = ok and bool(np.array_equal(mi, wi) and np.array_equal(ms, ws))
    q.put((rank, ok, x.world, x.is_rccl))
    x.close()


def test_cpp_exchange_binding_world3_tcp():
    """bench.py --exchange cpp: libafis_exchange.so's afis_exchange_* entry points (the `match` host's own exchange step) from Python,
    three ranks, over the TCP stand-in — same merged lists as the global top-k."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpp_exchange_worker, args=(r, 3, port - 1, q)) for r in (2, 1, 0)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert sorted(res) == [(0, True, 3, False), (1, True, 3, False), (2, True, 3, False)]


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` (the driver's form) is a launcher: with fewer than N devices it must fail loudly, never run one rank."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(2, n + 1))], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "does not match --gpus" in r.stderr


def test_bench_launcher_tears_the_job_down_when_a_rank_dies():
    """`python bench.py --gpus 4` (the driver's form: bench.py is its own launcher) with rank 2 dying before the rendezvous: the launcher polls EVERY
    child, so it reports that rank's exit code and stops the three others — which sit in the rendezvous waiting for it — within seconds instead of
    waiting on rank 0 until the backend's own timeout.  (No GPU needed: the ranks never get as far as creating a matcher.)"""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--share-gpu", "--backend", "gloo", "--gallery", "200", "--queries", "1"],
                       capture_output=True, text=True, timeout=280, env=dict(env, AFIS_BENCH_FAIL_RANK="2"))
    assert r.returncode == 7, (r.returncode, r.stderr[-1500:])
    assert "rank 2 exited with code 7" in r.stderr
    assert time.time() - t0 < 120
