"""N > 1 end to end on ONE GPU: two ranks share GPU 0, each owns one gallery shard, the exchange step and the merge run for real.

* bench.py --gpus 2 --share-gpu --backend gloo: the Python host path (torch.distributed all_gather of the per-shard top-24 lists).
* match with WORLD_SIZE=2 and AFIS_EXCHANGE=tcp: the C++ host path — shard planning, the padded score-column gather of -ldir, the
  top-24 merge and the correspondence-file ownership of -l — with the TCP gather standing in for RCCL (two ranks cannot share a GPU
  under RCCL).  RCCL stays the production exchange; on a 1-GPU box it is exercised with a communicator of one rank
  (test_cli_exchange_path_single_rank).
Rank lists / files must equal the single-process run bit for bit."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")
M = importlib.import_module("msu-latentafis_amd.host.matcher")


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_bench_two_ranks_share_one_gpu(tmp_path):
    common = ["--gallery", "3000", "--queries", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--dump-ranks", str(tmp_path / "one.npz")],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                          *common, "--dump-ranks", str(tmp_path / "two.npz")], capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    import json
    j2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert j2["n_gpus"] == 2 and j2["rank1_hits"] == "6/6" and j2["config"]["parallelism"] == "gallery-shard x2"
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["score"], b["score"])
    # the line certifies who ran: two ranks, ONE device (the same PCI bus id and UUID twice), no RCCL communicator (gloo) — it cannot be mistaken for a two-GPU run
    assert len(j2["ranks"]) == 2 and [r["rank"] for r in j2["ranks"]] == [0, 1]
    assert j2["distinct_devices"] == 1 and j2["shared_gpu"] is True and j2["rccl_ranks"] == 0
    assert j2["ranks"][0]["pci_bus_id"] == j2["ranks"][1]["pci_bus_id"] != "" and j2["ranks"][0]["uuid"] == j2["ranks"][1]["uuid"]
    assert j2["ranks"][0]["shard"][1] == j2["ranks"][1]["shard"][0] and j2["ranks"][1]["shard"][1] == 3000
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert j1["distinct_devices"] == 1 and j1["shared_gpu"] is False and j1["rccl_ranks"] == 0 and len(j1["ranks"]) == 1 and j1["ranks"][0]["compute_units"] == 256


def test_one_rank_with_the_exchange_step_equals_the_plain_run(tmp_path):
    """The N = 1 invariant: `bench.py --gpus 1 --force-dist` (a process group of one rank, the RCCL all-gather of the rank lists executed for real — torch's and the C++
    host's own) must print the rank lists of the plain single-process run, and its line must say: one device, one RCCL rank."""
    import json
    common = ["--gallery", "3000", "--queries", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--dump-ranks", str(tmp_path / "one.npz")], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = np.load(tmp_path / "one.npz")
    for exch in ("torch", "cpp"):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), AFIS_EXCHANGE_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        env.pop("AFIS_EXCHANGE", None)
        f = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--exchange", exch, *common, "--dump-ranks", str(tmp_path / f"f_{exch}.npz")],
                           capture_output=True, text=True, timeout=600, env=env)
        assert f.returncode == 0, f.stderr[-2000:]
        j = json.loads(f.stdout.strip().splitlines()[-1])
        b = np.load(tmp_path / f"f_{exch}.npz")
        assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["score"], b["score"]), exch
        assert j["n_gpus"] == 1 and j["distinct_devices"] == 1 and j["shared_gpu"] is False and j["rccl_ranks"] == 1, (exch, j["rccl_ranks"], j["rccl_ranks_is"])
        if exch == "cpp": assert j["ranks"][0]["rccl_comm_count"] == 1 and j["ranks"][0]["rccl_comm_device"] == 0


def _run_match(exe, args, cwd, world=1, extra_env=None, timeout=300):
    """Start `world` ranks of match on GPU 0; returns [(returncode, stdout, stderr)] in rank order."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        if world > 1:
            env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       AFIS_EXCHANGE="tcp", AFIS_EXCHANGE_TIMEOUT_S="60")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([exe, *args, "-d", "0"], cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        out.append((p.returncode, o, e))
    return out


@pytest.fixture(scope="module")
def match_case(tmp_path_factory, codebook_bytes):
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(M.LIB_PATH), "match"], check=True)
    cb = T.Codebook.from_bytes(codebook_bytes)
    lats, gal = cases.small_set(cb, seed=5, n_lat=3, n_gal=21)
    d = tmp_path_factory.mktemp("mr")
    (d / "work").mkdir(); (d / "gal").mkdir(); (d / "lat").mkdir()
    (d / "cb.dat").write_bytes(codebook_bytes)
    for j, g in enumerate(gal):
        (d / "gal" / f"R{j:03d}.dat").write_bytes(T.write_rolled(g))
    (d / "gal" / "R_empty.dat").write_bytes(b"")
    for i, L in enumerate(lats):
        (d / "lat" / f"L{i}.dat").write_bytes(T.write_latent(L))
    return exe, d


def _files(path):
    return {f: open(os.path.join(path, f), "rb").read() for f in sorted(os.listdir(path))}


def test_match_two_ranks_equal_one_rank(match_case):
    exe, d = match_case
    # (-tie 2: option ref_tie_order and, for -l, the rank list by std::sort on the gathered score column — every rank sorts the same whole column)
    for mode, extra in (("ldir", ["-ldir", str(d / "lat")]), ("l", ["-l", str(d / "lat" / "L0.dat")]), ("ltie", ["-l", str(d / "lat" / "L0.dat"), "-tie", "2"]), ("ldirtie", ["-ldir", str(d / "lat"), "-tie", "2"])):
        outs = {}
        for world in (1, 2):
            sd = d / f"out_{mode}_{world}"
            sd.mkdir()
            res = _run_match(exe, [*extra, "-g", str(d / "gal"), "-s", str(sd) + "/", "-c", str(d / "cb.dat")], d / "work", world)
            for rc, o, e in res:
                assert rc == 0, (mode, world, e[-1500:])
            if world == 2:
                assert res[1][1] == ""                                   # rank 1 is silent on stdout; rank 0 speaks for the job
                assert "Gallery size: 22" in res[0][1]
            outs[world] = _files(sd)
        assert set(outs[1]) == set(outs[2]) and len(outs[1]) >= (3 if mode.startswith("ldir") else 2), (mode, sorted(outs[1]), sorted(outs[2]))
        for f in outs[1]:
            assert outs[1][f] == outs[2][f], (mode, f)
        if mode in ("l", "ltie"):
            assert any(f.startswith("corrL0_") and len(outs[1][f]) > 0 for f in outs[1])      # correspondence files came from both shards' owners


def test_match_rank_failure_stops_the_whole_job(match_case, tmp_path):
    """One rank's shard holds a rolled .dat the C ABI rejects (minutiae descriptor length 95): that rank fails in load_gallery,
    the agreement point makes BOTH ranks leave with a non-zero status instead of one waiting in the collective for ever."""
    exe, d = match_case
    import shutil
    g2 = tmp_path / "gal"
    shutil.copytree(d / "gal", g2)
    rng = np.random.default_rng(3)
    bad = T.FPTemplate(minu=[T.MinutiaeTemplate(np.arange(5, dtype=np.int16), np.arange(5, dtype=np.int16), np.zeros(5, np.float32),
                                                rng.standard_normal((5, 95)).astype(np.float32))], tex=[])
    files = sorted(os.listdir(g2))
    # directory order is what the ranks shard by: overwrite EVERY file of the second half so the bad shard is rank 1's whatever the order
    listing = subprocess.run([exe, "-ldir", str(d / "lat"), "-g", str(g2), "-s", str(tmp_path / "o0") + "/", "-c", str(d / "cb.dat"), "-d", "0"],
                             cwd=d / "work", capture_output=True, text=True, timeout=300)
    assert listing.returncode == 0
    order = [l.split('"')[1] for l in listing.stdout.splitlines() if l.startswith("rolled template file")]
    assert len(order) == len(files)
    open(order[-1], "wb").write(T.write_rolled(bad))                    # the last template of the listing: rank 1's shard
    (tmp_path / "o2").mkdir()
    res = _run_match(exe, ["-ldir", str(d / "lat"), "-g", str(g2), "-s", str(tmp_path / "o2") + "/", "-c", str(d / "cb.dat")], d / "work", 2, timeout=120)
    assert res[0][0] != 0 and res[1][0] != 0, res
    assert "des_len must be 96" in res[1][2] and "another rank failed" in res[0][2], (res[0][2][-500:], res[1][2][-500:])


def test_bench_driver_command_form_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher — the form the round driver uses — must start two ranks by itself, report
    n_gpus = 2 and give the 1-rank run's rank lists."""
    import json
    common = ["--gallery", "3000", "--queries", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--dump-ranks", str(tmp_path / "one.npz")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    for exchange in ("torch", "cpp"):
        two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--exchange", exchange,
                              *common, "--dump-ranks", str(tmp_path / f"two_{exchange}.npz")], capture_output=True, text=True, timeout=900,
                             env=dict(env, AFIS_EXCHANGE="tcp"))
        assert two.returncode == 0, two.stderr[-2000:]
        lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, two.stdout[-1000:]                       # ONE JSON line, from rank 0
        j2 = json.loads(lines[0])
        assert j2["n_gpus"] == 2 and j2["rank1_hits"] == "6/6" and j2["config"]["parallelism"] == "gallery-shard x2" and j2["config"]["exchange"].startswith(exchange)
        a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / f"two_{exchange}.npz")
        assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["score"], b["score"]), exchange
    # asking for more GPUs than the box has is an error, not a silent 1-GPU run
    import torch
    n = torch.cuda.device_count()
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(2, n + 1)), *common], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and "GPU(s) visible" in bad.stderr



def test_eight_rank_rehearsal_of_the_driver_command(tmp_path):
    """The command the round driver runs on an 8-GPU node — `python bench.py --gpus 8` with no launcher — rehearsed with all eight ranks on GPU 0
    (eight gallery shards, eight processes, the exchange step between eight ranks): ONE JSON line, n_gpus == 8, the merged rank lists equal to the
    one-rank run's, with the Python exchange (torch.distributed) and with the C++ host's own (csrc/rank_exchange.cpp, its TCP form: RCCL cannot put
    two ranks on one GPU).  No scaling number comes out of this: it proves the 8-rank path, not its speed."""
    import json
    common = ["--gallery", "3000", "--queries", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--dump-ranks", str(tmp_path / "one.npz")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    a = np.load(tmp_path / "one.npz")
    for exchange in ("cpp", "torch"):
        r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--backend", "gloo", "--exchange", exchange,
                             *common, "--dump-ranks", str(tmp_path / f"eight_{exchange}.npz")], capture_output=True, text=True, timeout=1500,
                            env=dict(env, AFIS_EXCHANGE="tcp"))
        assert r8.returncode == 0, r8.stderr[-2000:]
        lines = [l for l in r8.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, r8.stdout[-1000:]
        j = json.loads(lines[0])
        assert j["n_gpus"] == 8 and j["rank1_hits"] == "6/6" and j["config"]["parallelism"] == "gallery-shard x8" and j["config"]["exchange"].startswith(exchange)
        pr = j["per_rank_ms_per_step"]
        assert 0 < pr["search"]["min"] <= pr["search"]["max"] and 0 <= j["exchange_ms_per_step"] <= pr["exchange_and_merge"]["max"]
        b = np.load(tmp_path / f"eight_{exchange}.npz")
        assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["score"], b["score"]), exchange


def test_match_eight_ranks_equal_one_rank(match_case):
    """`match` started eight times (WORLD_SIZE=8, AFIS_EXCHANGE=tcp, every rank on GPU 0): 22 templates in eight shards of 2-3, -ldir and -l;
    every score / rank / correspondence file byte-equal to the single-process run."""
    exe, d = match_case
    for mode, extra in (("ldir", ["-ldir", str(d / "lat")]), ("l", ["-l", str(d / "lat" / "L0.dat")])):
        outs = {}
        for world in (1, 8):
            sd = d / f"out8_{mode}_{world}"
            sd.mkdir()
            res = _run_match(exe, [*extra, "-g", str(d / "gal"), "-s", str(sd) + "/", "-c", str(d / "cb.dat")], d / "work", world, timeout=600)
            for rc, o, e in res:
                assert rc == 0, (mode, world, e[-1500:])
            if world == 8:
                assert all(o == "" for _, o, _ in res[1:]) and "Gallery size: 22" in res[0][1]
            outs[world] = _files(sd)
        assert set(outs[1]) == set(outs[8]) and len(outs[1]) >= (3 if mode == "ldir" else 2), (mode, sorted(outs[1]), sorted(outs[8]))
        for f in outs[1]:
            assert outs[1][f] == outs[8][f], (mode, f)
