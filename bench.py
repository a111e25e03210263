#!/usr/bin/env python
"""bench.py — latent queries/s against a resident rolled gallery on N MI355X (one process per GPU).

A "step" = one pass of the hot path: Q latents scored against the whole G-template gallery (LUT build, PQ-ADC row-max,
texture tail, minutiae scorer, fusion, per-shard top-24) plus the one exchange step (RCCL all_gather of the per-shard
rank lists) and the merge.  Gallery and queries are resident in HBM before the timed region; scores/rank lists come back
to the host inside it.  Default workload = BASELINE.json configs[2]: batch 100 latents vs a 100k synthetic gallery.
The gallery is FIXED at G as N grows (gallery shards across ranks): "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")
M = importlib.import_module("msu-latentafis_amd.host.matcher")
SH = importlib.import_module("msu-latentafis_amd.host.sharding")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (2065 TF measured with tools/ubench/mfma_layout.hip, one wave per SIMD)
LDS_PEAK_BYTES = 256 * 256 * 2.4e9     # 256 CUs x 256 B/clk (ds_read_b128) x 2.4 GHz = 157 TB/s (MI355X_MICROARCH.md §LDS); 154.8 TB/s measured with
                                      # tools/ubench/lds_rate.hip (profiles/r01_lds_peak.json)
BYTES_PER_TEX_POINT = 2 + 2 + 4 + 16  # x, y, ori, 16 PQ code bytes per rolled texture point
BYTES_PER_MINUTIA = 2 + 2 + 4 + 96 * 4


def physical_cores():
    """(physical cores, logical CPUs) of this host from /proc/cpuinfo (threads != cores on an SMT host)."""
    try:
        ids, n = set(), 0
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"): n += 1
            elif line.startswith("physical id"): phys = line.split(":")[1].strip()
            elif line.startswith("core id"): core = line.split(":")[1].strip(); ids.add((phys, core))
        return (len(ids) or n), n
    except Exception:
        return os.cpu_count() or 1, os.cpu_count() or 1


def host_cpu_limits():
    """What this process may actually use: the affinity mask and the cgroup CPU quota (a container can see 256 logical CPUs and own 16)."""
    info = {"affinity_cpus": None, "cgroup_cpu_max": None, "cgroup_quota_cpus": None}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for pth in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(pth).read().strip()
            info["cgroup_cpu_max"] = txt
            if pth.endswith("cpu.max"):
                q, per = txt.split()
                if q != "max": info["cgroup_quota_cpus"] = round(float(q) / float(per), 2)
            else:
                q = float(txt)
                if q > 0: info["cgroup_quota_cpus"] = round(q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
            break
        except Exception:
            continue
    return info


def cpu_baseline(cb_bytes, lats, gal, lo, pairs_per_thread=800):
    """The CPU restatement of the reference path (oracle/, OpenMP) timed on bounded samples of the same workload:
       compute-only   gallery parsed once and resident in RAM, T host threads taking one pair at a time (dynamic schedule), for T = 8, 16, 32, ...
                      up to every CPU this process may use: the CURVE is reported, the best point is the `value` of cpu_baseline.  Every point scores
                      pairs_per_thread x T pairs — about 2 s of wall time at the ~350 pairs/s a core manages (round 3 timed 55 ms per point: noise-limited)
       compute-only   8 threads, schedule(static,16): the reference's OpenMP setting (matcher.cpp:168, :273)
       reference-faithful  8 threads, static 16, and every rolled .dat RE-READ AND RE-PARSED for every pair from page-cache-warm
                      files, which is what the reference's loop does (matcher.cpp:173, :278)."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle()
    ocb = orc.codebook(cb_bytes)
    lim = host_cpu_limits()
    usable = lim["affinity_cpus"] or orc.lib.orc_num_threads()
    if lim["cgroup_quota_cpus"]: usable = max(1, min(usable, int(lim["cgroup_quota_cpus"] + 0.5)))
    ladder = sorted({t for t in (8, 16, 32, 64, 128, 256) if t < usable} | {usable})
    hl = orc.latent(ocb, T.write_latent(lats[0]))[0]
    n_max = int(min(gal.G, max(480, pairs_per_thread * min(ladder[-1], 32))))    # at most 25 600 templates parsed for the sample
    dats = [T.write_rolled(gal.template(g)) for g in range(n_max)]
    hr = [orc.rolled(d)[0] for d in dats]
    orc.search(ocb, hl, hr[:64], tie_mode=1, threads=min(usable, 64))                 # warm: page in the LUT and the code
    curve = []
    for t in ladder:
        n = min(n_max, max(pairs_per_thread * t, 160))
        dt = 1e30
        for _ in range(2):                                                 # the first call at a new thread count also starts the OpenMP team: keep the better of two
            t0 = time.perf_counter(); orc.search(ocb, hl, hr[:n], tie_mode=1, threads=t); dt = min(dt, time.perf_counter() - t0)
        curve.append({"threads": t, "pairs": n, "pairs_per_s": round(n / dt, 1)})
    best = max(curve, key=lambda c: c["pairs_per_s"])
    n8 = min(len(hr), max(480, pairs_per_thread * 8))
    wall8 = 1e30
    for _ in range(2):
        t1 = time.perf_counter(); orc.search(ocb, hl, hr[:n8], tie_mode=1, threads=0); wall8 = min(wall8, time.perf_counter() - t1)
    # reference-faithful leg: the same n8 templates as files in a tmpdir (written and read once: page-cache warm)
    tmp = tempfile.mkdtemp(prefix="afis_cpu_baseline_")
    try:
        paths = []
        for g in range(n8):
            pth = os.path.join(tmp, "R%06d.dat" % g)
            with open(pth, "wb") as f: f.write(dats[g])
            with open(pth, "rb") as f: f.read()
            paths.append(pth)
        t2 = time.perf_counter(); _, sf = orc.search_files(ocb, hl, paths, tie_mode=1, threads=0); wallf = time.perf_counter() - t2
        _, sr = orc.search(ocb, hl, hr[:n8], tie_mode=1, threads=0)
        assert np.array_equal(sf, sr)                                    # same scores either way; only the time differs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"pairs_per_s": best["pairs_per_s"], "threads": best["threads"], "curve": curve, "limits": lim, "usable_cpus": usable,
            "sample": f"1 latent x {best['pairs']} gallery templates (templates {lo}..{lo + best['pairs']} of the bench gallery), {best['threads']} threads, one pair at a time (dynamic schedule)",
            "pairs_per_s_8_threads": n8 / wall8, "pairs_per_s_reference_faithful": n8 / wallf,
            "sample_8_threads": f"1 latent x {n8} gallery templates, 8 threads schedule(static,16)"}


class PowerSampler:
    """Board power of one GPU while a region runs, so that a schedule or a kernel change can be judged in joules as well as in milliseconds (the bound pass is power-limited:
    DESIGN section 8).  Source: the amdgpu hwmon file of the device (power1_average or power1_input, microwatts; the card whose PCI bus id matches), read by a thread every 50 ms;
    when the driver exposes none, `rocm-smi --showpower --json` once a second.  Returns None when neither works."""

    def __init__(self, pci_bus_id):
        import glob
        self.path = None; self.samples = []; self._stop = False; self._thr = None; self.source = None
        want = (pci_bus_id or "").lower()
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                bus = os.path.basename(os.path.realpath(dev)).lower()
            except OSError:
                continue
            if want and bus != want: continue
            for name in ("power1_average", "power1_input"):
                hits = glob.glob(os.path.join(dev, "hwmon", "hwmon*", name))
                if hits: self.path = hits[0]; self.source = "hwmon " + name; break
            if self.path: break
        if not self.path:
            import shutil
            self.smi = shutil.which("rocm-smi")
            if self.smi: self.source = "rocm-smi --showpower"
        else:
            self.smi = None

    def _read(self):
        if self.path:
            return int(open(self.path).read().strip()) * 1e-6
        import subprocess
        o = subprocess.run([self.smi, "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        vals = [float(v) for card in json.loads(o).values() for k_, v in card.items() if "ower" in k_ and str(v).replace(".", "", 1).isdigit()]
        return vals[0] if vals else None

    def start(self):
        if not (self.path or self.smi): return self
        import threading
        self.samples = []; self._stop = False
        def run():
            while not self._stop:
                try:
                    w = self._read()
                    if w: self.samples.append(w)
                except Exception:
                    pass
                time.sleep(0.05 if self.path else 1.0)
        self._thr = threading.Thread(target=run, daemon=True); self._thr.start()
        return self

    def stop(self, seconds, units):
        """-> {"watts_mean", "watts_max", "samples", "joules_per_query", "source"} or None"""
        self._stop = True
        if self._thr: self._thr.join(timeout=15)
        if not self.samples: return None
        w = float(np.mean(self.samples))
        return {"watts_mean": round(w, 1), "watts_max": round(float(np.max(self.samples)), 1), "samples": len(self.samples), "joules_per_query": round(w * seconds / max(1, units), 3), "source": self.source}


def _free_ports(n):
    """n distinct TCP ports that are free on 127.0.0.1 right now (bound together, then released)."""
    import socket
    socks = []
    for _ in range(n):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); socks.append(sk)
    ports = [sk.getsockname()[1] for sk in socks]
    for sk in socks: sk.close()
    return ports


def spawn_ranks(n):
    """`python bench.py --gpus N` started by hand or by the driver (no torchrun): become the launcher — N copies of this command line, one
    process per GPU (LOCAL_RANK = GPU), verified-free rendezvous ports on 127.0.0.1 (torch's store, and the C++ exchange's own id hand-off);
    rank 0 owns stdout (the one JSON line).  EVERY child is polled: the first non-zero exit tears the whole job down at once (a rank that waits
    in a collective for a dead peer would otherwise sit there until the backend's own timeout)."""
    import subprocess
    port, xport = _free_ports(2)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AFIS_EXCHANGE_PORT=str(xport), AFIS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))

    def tear_down():
        for q in procs:
            if q.poll() is None: q.terminate()
        t_end = time.time() + 10
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()

    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = bad[0][1]
                print(f"bench.py: rank {bad[0][0]} exited with code {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                tear_down()
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        tear_down()
        rc = 130
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gallery", type=int, default=100000)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--workload", default="headline", choices=sorted(S.WORKLOADS), help="headline = BASELINE.json's shapes (rolled minutiae clip(N(80, 15), 20, 200), latent U{20..60}); "
                    "wide = the shapes the reference's reader also accepts but the synthetic envelope never produced (rolled clip(N(130, 40), 20, 400), latent U{20..150}): not the headline")
    ap.add_argument("--dup", type=int, default=10, choices=[0, 10, 30], help="--workload structured: the share (%%) of a rolled template's texture points whose 16-byte code vector also occurs at another point of the template")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--k", type=int, default=24)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--query-batch", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--tile-share", type=int, default=0)
    ap.add_argument("--bound-cus", type=int, default=-1, help="CUs the bound pass is confined to, the minutiae stage running beside it on the others (-1 = the library's default, 128; 0 = one stream, kernels back to back)")
    ap.add_argument("--s3-tie-order", type=int, default=0, choices=[0, 1, 2], help="option s3_tie_order: 1 = candidate norms that tie in libstdc++'s std::sort order, as the reference binary (lists short of 120 positive similarities go through the any-shape kernel); NOT the headline setting")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--score-stats", action="store_true", help="after the timed region: the share of non-mate pairs with a positive score (always reported for --workload structured)")
    ap.add_argument("--no-alone", action="store_true", help="skip the extra back-to-back steps after the timed region (roofline.alone_on_the_chip): for profiler runs, whose per-kernel averages they would mix into")
    ap.add_argument("--refine-stats", action="store_true", help="adc_variant 9: report what the selection / exact-recomputation kernel did (one extra step after the timed region, in a second context on the test library)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the rank-list exchange (nccl = RCCL; gloo for tests)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "cpp"], help="the rank-list exchange step: cpp = the `match` host's own exchange (csrc/rank_exchange.cpp: ONE ncclAllGather of the "
                    "(index, score) block per step, or its TCP stand-in with AFIS_EXCHANGE=tcp) through libafis_exchange.so; torch = torch.distributed all_gather (host/sharding.py); "
                    "auto (default) = cpp when there is more than one rank and the backend is nccl (the timed step then contains the repository's own collective), torch otherwise")
    ap.add_argument("--share-gpu", action="store_true", help="test mode: every rank uses GPU 0 (validates the N>1 path on a 1-GPU box)")
    ap.add_argument("--dump-ranks", default="", help="rank 0 writes the merged rank lists of the last step to this .npz (tests)")
    ap.add_argument("--force-dist", action="store_true", help="test mode: initialise torch.distributed and run the exchange step even when WORLD_SIZE is 1")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    if a.gpus < 1: sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:                     # the driver's form: python bench.py --gpus N (no launcher)
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus and not a.share_gpu:
            sys.exit(f"bench.py: --gpus {a.gpus} but only {n_dev} GPU(s) visible (use --share-gpu to put every rank on GPU 0: test mode)")
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AFIS_BENCH_FAIL_RANK") == str(rank) and world > 1:          # test hook (tests/test_sharding.py): this rank dies before the rendezvous
        sys.exit(7)
    if world != a.gpus and not a.force_dist:
        sys.exit(f"bench.py: WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if world > 1 and not a.share_gpu and torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} visible")
    gpu = 0 if a.share_gpu else local
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":       # RCCL's version banner goes to stdout; keep stdout to the one JSON line
            del os.environ["NCCL_DEBUG"]
        if a.backend == "nccl":
            torch.cuda.set_device(gpu)
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        else:                                                           # gloo (tests): rendezvous first — nothing a rank can fail on sits between the test hook above and the point where the ranks wait for each other
            dist.init_process_group(a.backend)
            torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu) if a.backend == "nccl" else torch.device("cpu")

    with open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb") as f:
        cb_bytes = f.read()
    cb = T.Codebook.from_bytes(cb_bytes)
    G, Q = a.gallery, a.queries

    # ---- synthetic workload (seeded; every rank generates only its own shard) -------------------------------------
    t_gen = time.perf_counter()
    wl = S.WORKLOADS[a.workload]
    nm_all, nt_all = S.gallery_counts(a.seed, G, **wl["gallery"])
    bounds = SH.shard_bounds(nt_all, world)                # balanced by rolled texture points, the cost driver
    lo, hi = bounds[rank]
    dup_measured = None
    if a.workload == "structured":
        # templates with the structure of extracted prints; the descriptors are drawn on the GPU (torch: plumbing) and encoded by the library's own encoder (afis_pq_encode)
        SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
        sigma = SS.DUP_SIGMA[a.dup]
        lats = SS.make_structured_latents(a.seed, Q, sigma=sigma, **wl["latent"])
        m_enc = M.Matcher(cb_bytes, device=gpu)
        gal = SS.make_packed_gallery_structured(a.seed, G, cb, lo, hi, sigma=sigma, encode=m_enc.pq_encode, device=torch.device("cuda", gpu), **wl["gallery"])
        m_enc.close()
        planted = SS.plant_structured_mates(a.seed, gal, cb, lats, G=G, lo=lo, sigma=sigma)
        n_s = min(gal.G, 2000)
        dup_measured = round(SS.dup_share(gal.tex_codes[:int(gal.tex_off[n_s])], gal.tex_off[:n_s + 1]), 4)
    else:
        lats = S.make_latents(a.seed, Q, **wl["latent"])
        gal = S.make_packed_gallery(a.seed, G, cb, lo, hi, **wl["gallery"])
        planted = S.plant_mates(a.seed, gal, cb, lats, G=G, lo=lo)
    t_gen = time.perf_counter() - t_gen

    m = M.Matcher(cb_bytes, device=gpu, taps=0 <= a.variant < 8)      # --variant 0..7 runs a reference kernel: libafis_hip_test.so
    if a.variant >= 0: m.set_option("adc_variant", a.variant)
    if a.query_batch > 0: m.set_option("query_batch", a.query_batch)
    if a.chunk > 0: m.set_option("chunk", a.chunk)
    if a.tile_share > 0: m.set_option("tile_share", a.tile_share)
    if a.bound_cus >= 0: m.set_option("bound_cus", a.bound_cus)
    if a.s3_tie_order: m.set_option("ref_tie_order", a.s3_tie_order)      # 1 = s3_tie_order 1; 2 = the greedy selections of S8 / S9 in std::sort's order as well
    bound_cus = m.get_option("bound_cus") if (a.variant < 0 or a.variant == 9) else 0
    if a.share_gpu and world > 1:                          # test mode: the ranks share one device, and each would otherwise budget its launch groups from all of its free memory
        m.set_option("rowmax_budget_mb", max(1024, int(0.5 * torch.cuda.mem_get_info(gpu)[1] / world) >> 20))
    t_up = time.perf_counter()
    m.gallery_add_packed(gal)
    m.gallery_commit(lo)
    qh = m.upload_queries(lats)
    t_up = time.perf_counter() - t_up

    exchange_note = None
    if a.exchange == "auto" and world > 1 and a.backend == "nccl":
        # the repository's own collective, unless its rendezvous (one TCP connection per rank on MASTER_PORT + 1 for the ncclUniqueId) fails on ANY rank: then every rank
        # uses torch.distributed instead (the ranks agree through a MIN all-reduce), and the line says so — the driver's 8-GPU run must not die of a busy port
        xch, why = None, ""
        try:
            xch = SH.CppExchange(gpu)
        except Exception as e_:                                             # noqa: BLE001
            why = str(e_)
        okt = torch.tensor([1 if xch is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 1:
            a.exchange = "cpp"
        else:
            if xch is not None: xch.close()
            xch = None; a.exchange = "torch"
            exchange_note = "auto: the C++ exchange could not be set up on every rank (%s); torch.distributed carries the exchange" % (why or "another rank failed")
    else:
        if a.exchange == "auto": a.exchange = "torch"
        xch = SH.CppExchange(gpu) if (a.exchange == "cpp" and use_dist) else None     # rendezvous on MASTER_PORT + 1 (torch's store owns MASTER_PORT)

    wall = {"search": 0.0, "exchange": 0.0}                 # host wall time of this rank's two halves of a step (reset after the warm-up)

    def step():
        t_a = time.perf_counter()
        r = m.search_resident(qh, k=a.k)                    # returns with the shard's rank lists on the host: the stream is drained
        t_b = time.perf_counter()
        if xch is not None:
            out_ = xch.gather_topk(r["topk_idx"], r["topk_score"], a.k)
        else:
            out_ = SH.gather_topk(r["topk_idx"], r["topk_score"], a.k, device=dev, force=a.force_dist)
        wall["search"] += t_b - t_a; wall["exchange"] += time.perf_counter() - t_b
        return out_

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    wall["search"] = wall["exchange"] = 0.0
    power = PowerSampler(m.device_info(gpu).get("pci_bus_id")).start() if rank == 0 else None
    t0 = time.perf_counter()
    tm_acc = None
    for _ in range(a.steps):
        idx, sc = step()
        tm = m.timing()
        tm_acc = tm if tm_acc is None else {k_: tm_acc[k_] + v for k_, v in tm.items()}
    sync()
    elapsed = time.perf_counter() - t0
    power_timed = power.stop(elapsed, Q * a.steps) if power is not None else None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1000.0 / max(1, a.steps)
    value = Q / (ms_per_step / 1000.0)
    if bound_cus > 0 and tm_acc.get("overlapped_groups", 1) == 0:            # the library ran every group back to back (a workload whose minutiae stage outweighs the bound pass): the pass had all CUs
        bound_cus_opt, bound_cus = bound_cus, 0
    else:
        bound_cus_opt = bound_cus
    # where a step's time goes on each rank: the search of the rank's shard, and the exchange step (all-gather + merge; a rank that finishes its
    # shard early waits here for the slowest one, so max(exchange) - min(exchange) is the imbalance and min(exchange) the cost of the step itself)
    per_rank = np.array([wall["search"], wall["exchange"]], np.float64) * 1000.0 / max(1, a.steps)
    pr_min, pr_max = per_rank.copy(), per_rank.copy()
    if use_dist:
        t_lo = torch.tensor(per_rank, dtype=torch.float64, device=dev); t_hi = t_lo.clone()
        dist.all_reduce(t_lo, op=dist.ReduceOp.MIN); dist.all_reduce(t_hi, op=dist.ReduceOp.MAX)
        pr_min, pr_max = t_lo.cpu().numpy(), t_hi.cpu().numpy()

    # ---- who ran: per rank the device the HIP runtime reports (name, PCI bus id, UUID), its shard and its wall times; rank 0 gathers them so that the line certifies
    # itself: N ranks on N DISTINCT devices with an N-rank RCCL communicator, or ranks sharing a GPU (test mode) ------------------------------------------------
    me = dict(m.device_info(gpu), rank=rank, local_rank=local, device_index=gpu, shard=[int(lo), int(hi)], search_ms=round(float(per_rank[0]), 3), exchange_ms=round(float(per_rank[1]), 3),
              rccl_comm_count=(xch.comm_count if xch is not None else None), rccl_comm_device=(xch.comm_device if xch is not None else None))
    ranks_info = [me]
    if use_dist:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)

    refine_stats = None
    if a.refine_stats and rank == 0:
        # a SECOND context on the test library (the counters are read through a parity tap) runs one step after the timed region: the counters cost five same-address atomics
        # per pair (round 6 measured +480 ms per step with them on), so they are never collected inside it
        m2 = M.Matcher(cb_bytes, device=gpu, taps=True)
        if a.variant >= 0: m2.set_option("adc_variant", a.variant)
        m2.set_option("mf_stats", 1); m2.set_option("rowmax_budget_mb", 16384)
        m2.gallery_add_packed(gal); m2.gallery_commit(lo)
        m2.search(lats, k=a.k, want_scores=False)
        refine_stats = m2.refine_stats()
        m2.close()
        rs_ = refine_stats
        refine_stats.update({"share_of_rows_evaluated": round(rs_["rows_evaluated"] / max(1, rs_["rows"]), 5), "cells_per_evaluated_row": round(rs_["cells_evaluated"] / max(1, rs_["rows_evaluated"]), 4),
                             "share_of_evaluated_rows_in_full": round(rs_["rows_evaluated_in_full"] / max(1, rs_["rows_evaluated"]), 6), "evaluated_rows_per_pair": round(rs_["rows_evaluated"] / max(1, rs_["pairs"]), 2)})
    # ---- rank-list sanity: the planted true mate must be rank 1 ------------------------------------------------------
    hits = sum(1 for q in range(Q) if int(idx[q, 0]) == planted[q][0][0])

    # ---- outside the timed region: the same step with the kernels back to back on one stream, so that the line also carries the bound pass's duration when it
    # has the whole chip (the roofline of the kernel by itself) and the per-stage times without overlap.  Two steps after one warm-up step; rank lists must not change.
    alone = None; power_alone = None
    if bound_cus > 0 and world == 1 and not a.no_alone:
        m.set_option("bound_cus", 0)
        m.search_resident(qh, k=a.k)
        acc = None
        power2 = PowerSampler(m.device_info(gpu).get("pci_bus_id")).start(); t_al = time.perf_counter()
        for _ in range(2):
            r_ = m.search_resident(qh, k=a.k); t_ = m.timing()
            acc = t_ if acc is None else {k_: acc[k_] + v for k_, v in t_.items()}
        power_alone = power2.stop(time.perf_counter() - t_al, Q * 2)
        assert np.array_equal(r_["topk_idx"], np.asarray(idx)) or use_dist, "the schedule changed a rank list"
        alone = {k_: (v / 2 if (k_.endswith("_ms") or k_.endswith("_ghz")) else v) for k_, v in acc.items()}
        m.set_option("bound_cus", bound_cus_opt)

    # ---- outside the timed region: the OTHER exchange of the path — `match -ldir` writes every score (matcher.cpp:201-204), so each rank contributes its shard's score
    # columns [Q][shard] (padded to the largest shard: 100 x 12 500 x 4 B = 5 MB per rank at 8 ranks) to one all-gather.  Three exchanges after one warm-up. ----
    exchange_ldir = None
    if use_dist:
        r_ = m.search_resident(qh, k=a.k, want_scores=True)
        widest = max(hi_ - lo_ for lo_, hi_ in bounds)
        blk = np.full((Q, widest), -1.0, np.float32); blk[:, :hi - lo] = r_["scores"]
        def ldir_exchange():
            if xch is not None: return xch.all_gather(blk)
            tb = torch.from_numpy(blk).to(dev); outl = [torch.empty_like(tb) for _ in range(world)]
            dist.all_gather(outl, tb); return torch.stack(outl).cpu().numpy()
        allb = ldir_exchange(); sync()
        t_l = time.perf_counter()
        for _ in range(3): allb = ldir_exchange()
        t_l = (time.perf_counter() - t_l) / 3
        full_scores = np.concatenate([allb[r__, :, :bounds[r__][1] - bounds[r__][0]] for r__ in range(world)], axis=1)      # [Q][G]: what rank 0 of `match -ldir` writes
        ok_top = all(int(np.lexsort((np.arange(G), -full_scores[q_].astype(np.float64)))[0]) == int(idx[q_, 0]) for q_ in range(Q))
        exchange_ldir = {"ms": round(t_l * 1000.0, 3), "bytes_per_rank": int(blk.nbytes), "what": "one all-gather of the per-shard score columns [Q][largest shard] f32 per rank, as `match -ldir` does (host buffers in, host buffers out: staging copies included)",
                         "rank1_of_the_gathered_scores_equals_the_rank_lists": bool(ok_top)}
        sync()
    # ---- outside the timed region: what share of the (latent, non-mate) pairs scores above zero (i.i.d. random templates: none; extracted prints: most) ----
    score_stats = None
    if world == 1 and (a.workload == "structured" or a.score_stats):
        r_ = m.search_resident(qh, k=a.k, want_scores=True, want_parts=True)
        sc_ = r_["scores"].copy(); pt_ = r_["parts"]
        mate = np.zeros_like(sc_, bool)
        for q_, lst in planted.items():
            for g_, _f in lst: mate[q_, g_ - lo] = True
        nm_ = ~mate & (sc_ >= 0)
        score_stats = {"non_mate_pairs": int(nm_.sum()), "non_mates_with_positive_score": round(float((sc_[nm_] > 0).mean()), 5),
                       "non_mates_with_positive_texture_score": round(float((pt_[..., 3][nm_] > 0).mean()), 5), "non_mates_with_positive_minutiae_score": round(float((pt_[..., :3][nm_] > 0).any(-1).mean()), 5),
                       "non_mate_score_mean": round(float(sc_[nm_].mean()), 4), "non_mate_score_max": round(float(sc_[nm_].max()), 3),
                       "true_mate_score_mean": round(float(np.mean([sc_[q_, planted[q_][0][0] - lo] for q_ in planted])), 2)}
    out = None
    if rank == 0 and a.dump_ranks:
        np.savez(a.dump_ranks, idx=np.asarray(idx), score=np.asarray(sc))
    if rank == 0:
        # roofline of the dominant kernel.  Durations come from HIP events on the stream the kernels run on (afis_get_timing).
        launches = max(1, tm_acc["adc_launches"])
        adc_ms_avg = tm_acc["adc_ms"] / launches
        q_per_launch = Q * a.steps / launches
        shard_tex_points = int(nt_all[lo:hi].sum())
        alg_bytes_launch = q_per_launch * shard_tex_points * BYTES_PER_TEX_POINT
        hbm_achieved = alg_bytes_launch / (adc_ms_avg * 1e-3) / 1e9 if adc_ms_avg > 0 else 0.0
        variant = 9 if a.variant < 0 else a.variant
        carried = None
        cp = next((p_ for p_ in (os.path.join(ROOT, "profiles", f) for f in ("r06_adc_counters.json", "r05_adc_counters.json", "r04_adc_counters.json")) if os.path.exists(p_)), "")
        carried_name = "profiles/" + os.path.basename(cp)
        if cp and world == 1 and G == 100000 and Q == 100 and variant == 9:   # measured for the default workload only (PMC passes with the kernels back to back, --bound-cus 0: a kernel's traffic does not depend on what runs beside it)
            try:
                carried = json.load(open(cp))
            except Exception:
                carried = None
        clock = {"bound_pass_ghz": round(tm_acc.get("bound_clock_ghz", 0.0) / a.steps, 4), "candidate_kernel_ghz": round(tm_acc.get("cands_clock_ghz", 0.0) / a.steps, 4),
                 "how": "in-kernel, this run: one lane of sampled workgroups (every 64th of the bound pass, the first 8 of the candidate kernel) reads s_memtime (shader cycles) and "
                        "s_memrealtime (100 MHz) at its first and last instruction; clock = sum of cycle differences / sum of tick differences x 0.1 GHz, averaged over the timed steps"}
        if variant == 9:
            # k_adc_mfma: every (latent texture row, rolled texture point) cell is a 96-long fp16 dot product on the matrix cores: 192 flop
            rows_per_step = sum(min(L.tex[0].n, 1000) if L.tex else 0 for L in lats)
            alg_flops_launch = rows_per_step * a.steps / launches * shard_tex_points * 192.0
            bound_ms_avg = tm_acc["adc_bound_ms"] / launches
            tflops = alg_flops_launch / (bound_ms_avg * 1e-3) / 1e12 if bound_ms_avg > 0 else 0.0
            share = (bound_cus / 256.0) if bound_cus > 0 else 1.0                   # the kernel is confined to this share of the chip's CUs (the rest runs the minutiae stage beside it)
            alone_ms = (alone["adc_bound_ms"] / max(1, alone["adc_launches"] / 2)) if alone else None
            alone_tf = (alg_flops_launch / (alone_ms * 1e-3) / 1e12) if alone else None
            ck_b, ck_c = clock["bound_pass_ghz"], clock["candidate_kernel_ghz"]
            if alone and alone.get("bound_clock_ghz", 0) > 0:
                limiting = ("measured in this run: alone on the chip the bound pass holds %.2f GHz against %.2f GHz under the candidate kernel (ratio %.2f) — the chip lowers its clock under this kernel's matrix work; "
                            "in the timed schedule (%s) it holds %.2f GHz. What the kernel's own loop can reach with tracking / decode compiled out is a row of profiles/r05_bound_pass_ablation.json / r04_bound_pass_ablation.json (timing-only ablations; not re-measured here)"
                            % (alone["bound_clock_ghz"], alone["cands_clock_ghz"], alone["bound_clock_ghz"] / max(1e-9, alone["cands_clock_ghz"]),
                               ("confined to %d CUs" % bound_cus) if bound_cus > 0 else "all CUs", ck_b))
            else:
                limiting = "measured in this run: %.2f GHz under the bound pass, %.2f GHz under the candidate kernel" % (ck_b, ck_c)
            roofline = {"bound": "mfma", "kernel": "k_adc_mfma (fp16 matrix-core bound pass over every cell; adc_variant 9)", "achieved": round(tflops, 2), "peak": MFMA_F16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tflops / MFMA_F16_PEAK_TFLOPS, 5),
                        "frac_is": "achieved / the WHOLE chip's fp16 matrix peak (2.5 PFLOP/s), whatever share of the CUs the kernel was given: one series with rounds 1-3 (round 4's line divided by the peak of the CUs used; that figure is frac_of_cus_used now)",
                        "cus_used": bound_cus if bound_cus > 0 else 256, "frac_of_cus_used": round(tflops / (MFMA_F16_PEAK_TFLOPS * share), 5),
                        "measured_clock_ghz": clock,
                        "alone_on_the_chip": ({"what": "the same kernel in the same process with the kernels back to back on one stream (bound_cus 0), two steps outside the timed region: its roofline when it has all 256 CUs",
                                               "avg_launch_ms": round(alone_ms, 3), "achieved": round(alone_tf, 2), "peak": MFMA_F16_PEAK_TFLOPS, "frac": round(alone_tf / MFMA_F16_PEAK_TFLOPS, 5),
                                               "measured_clock_ghz": {"bound_pass_ghz": round(alone.get("bound_clock_ghz", 0.0), 4), "candidate_kernel_ghz": round(alone.get("cands_clock_ghz", 0.0), 4)}} if alone else None),
                        "traffic": carried.get("traffic_bytes_per_launch") if carried else None,
                        "traffic_source": (carried_name + " (carried, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload; FETCH_SIZE x 2 for the 16 B/lane streams as "
                                           "MI355X_MICROARCH.md prescribes and profiles/r03_fetch_calibration.json confirms; collected with the kernels back to back, --bound-cus 0; PMC counters cannot be read inside this run)") if carried else None,
                        "stage_traffic": carried.get("stage_traffic") if carried else None,
                        "achieved_is": "ALGORITHMIC flops (latent texture rows of the launch x rolled texture points of the shard x 192) / average kernel duration; padding rows / points and the "
                                       "recomputation kernel's work are not counted",
                        "alg_flops_per_launch": alg_flops_launch, "avg_launch_ms": round(bound_ms_avg, 3),
                        "refine_kernel_avg_launch_ms": round(tm_acc["adc_refine_ms"] / launches, 3),
                        "hbm_view": {"what": "the same stage (bound pass + recomputation) priced as north_star prices it: 24 algorithmic bytes per rolled texture point per query / stage time", "alg_bytes_per_launch": alg_bytes_launch,
                                     "achieved_GBps": round(hbm_achieved, 2), "peak_GBps": HBM_PEAK_GBS, "frac": round(hbm_achieved / HBM_PEAK_GBS, 6)},
                        "unit_fractions_from_counters": carried.get("fractions") if carried else None,
                        "unit_fractions_source": (carried_name + " (carried)") if carried else None,
                        "limiting_resource": limiting}
        else:
            lookups_per_s = tm_acc["adc_lookups"] / (tm_acc["adc_ms"] * 1e-3) if tm_acc["adc_ms"] > 0 else 0.0
            quantised = variant == 8                                            # the 16-bit pass: 2 LDS bytes per look-up
            lds_bytes_per_lookup = 2 if quantised else 4
            roofline = {"bound": "hbm", "kernel": "k_adc_rowmin_q<1024,true> (16-bit LDS-table bound pass + exact refine)" if variant == 8 else "k_adc_rowmax (direct exact kernel)",
                        "achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 6), "traffic": None, "traffic_source": None,
                        "achieved_is": "ALGORITHMIC bytes (24 B per rolled texture point per query of the launch) / kernel time; by construction not the binding resource: see lds_frac and limiting_resource",
                        "alg_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": round(adc_ms_avg, 3),
                        "lds_lookups_per_s": lookups_per_s, "lds_bytes_per_lookup": lds_bytes_per_lookup, "lds_peak_bytes_per_s": LDS_PEAK_BYTES,
                        "lds_frac": round(lookups_per_s * lds_bytes_per_lookup / LDS_PEAK_BYTES, 4),
                        "limiting_resource": "vector instruction issue (profiles/r02_pmc_sq_summary.txt: 8.27e10 VALU instructions per launch at 3.5 cycles each, profiles/r03_valu_peak.json, = 0.78 of the SIMD-cycles); LDS array 0.55 busy"}
        pipeline_bytes = Q * (int(nt_all.sum()) * BYTES_PER_TEX_POINT + int(nm_all.sum()) * BYTES_PER_MINUTIA)
        out = {
            "metric": "latent queries/sec vs 100k rolled gallery", "value": round(value, 4), "unit": "queries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch {Q} latents vs {G}-template synthetic rolled gallery "
                                   f"({'BASELINE.json configs[2]' if (Q, G, a.workload) == (100, 100000, 'headline') else 'NOT the headline: ' + ('workload wide — rolled minutiae clip(N(130, 40), 20, 400), latent minutiae U{20..150}, everything else as configs[2]' if a.workload == 'wide' else ('workload structured — configs[2] sizes; texture points on the 16-px grid inside a foreground blob, smooth ridge flow, descriptors near a shared manifold PQ-encoded afterwards, --dup %d' % a.dup) if a.workload == 'structured' else 'not the headline size')}); planted mates; top-{a.k} rank lists",
                       "workload_name": a.workload, "structured_dup_share_measured": dup_measured, "queries": Q, "gallery": G, "parallelism": f"gallery-shard x{world}",
                       "exchange": ("none (one rank)" if not use_dist else
                                    ("cpp: csrc/rank_exchange.cpp " + ("ncclAllGather (RCCL)" if xch.is_rccl else "TCP stand-in (AFIS_EXCHANGE=tcp)")) if xch is not None
                                    else f"torch: torch.distributed all_gather, backend {a.backend}"),
                       "exchange_note": exchange_note, "s3_tie_order": a.s3_tie_order, "adc_variant": variant, "bound_cus": bound_cus, "bound_cus_option": bound_cus_opt,
                       "schedule": ("bound pass on %d CUs, minutiae stage (candidates + lists) beside it on the other %d, then recomputation + texture lists on the whole chip; stage times overlap: their sum exceeds ms_per_step" % (bound_cus, 256 - bound_cus))
                                   if bound_cus > 0 else
                                   ("one stream, the kernels of a launch group back to back" + (" (the library chose it for this workload: the minutiae stage outweighs the bound pass, option bound_cus %d notwithstanding)" % bound_cus_opt if bound_cus_opt > 0 else "")),
                       "overlapped_groups_per_step": tm_acc.get("overlapped_groups", 0) // max(1, a.steps), "bound_pass_dtype": ("f16 operands, f32 accumulation on the matrix cores: used for BOUNDS only, every score is the reference's f32 arithmetic" if variant == 9 else
                                                                   "u16 fixed point in LDS: used for BOUNDS only" if variant == 8 else "none"), "mean_latent_tex_rows": float(np.mean([L.tex[0].n for L in lats])),
                       "mean_rolled_tex_points": float(nt_all.mean()), "mean_rolled_minutiae": float(nm_all.mean()),
                       "mean_latent_minutiae_selected": float(np.mean([L.minu[i].n for L in lats for i in (26, 2, 11) if len(L.minu) > i]))},
            "roofline": dict(roofline, pipeline_achieved_GBps=round(pipeline_bytes / (ms_per_step * 1e-3) / 1e9, 3)),
            "stage_ms_per_step": {k_: round(tm_acc[k_] / a.steps, 3) for k_ in ("lut_ms", "adc_ms", "adc_bound_ms", "adc_refine_ms", "tex_tail_ms", "minu_ms", "cands_ms", "minu_graph_ms", "fuse_ms", "topk_ms", "total_ms")},
            "stage_ms_per_step_back_to_back": ({k_: round(alone[k_], 3) for k_ in ("lut_ms", "adc_ms", "adc_bound_ms", "adc_refine_ms", "tex_tail_ms", "minu_ms", "cands_ms", "minu_graph_ms", "fuse_ms", "topk_ms", "total_ms")} if alone else None),
            "minutiae_candidate_tasks": {"per_step": int(tm_acc.get("minu_tasks", 0) // a.steps),
                                         "fast_kernel_small_class": int(tm_acc.get("minu_tasks_small", 0) // a.steps), "fast_kernel_medium_class": int(tm_acc.get("minu_tasks_medium", 0) // a.steps),
                                         "fast_kernel_large_class": int(tm_acc.get("minu_tasks_large", 0) // a.steps), "any_shape_fallback_kernel": int(tm_acc.get("minu_fallback_tasks", 0) // a.steps),
                                         "fallback_share": round(tm_acc.get("minu_fallback_tasks", 0) / max(1, tm_acc.get("minu_tasks", 0)), 6),
                                         "fast_kernel_limits": {k_: m.get_option(k_) for k_ in ("minu_fast_max_latent", "minu_fast_max_rolled", "minu_fast_max_cells")},
                                         "how": "counted by the kernels of this run (afis_timing.minu_*): small = <= 64 x 128 minutiae (256-thread workgroups), medium = <= 16 384 similarities (512), large = <= 512 rolled minutiae and <= 38 912 similarities incl. the stride padding (1024)"},
            "refine_stats": refine_stats, "score_stats": score_stats,
            "per_rank_ms_per_step": {"search": {"min": round(float(pr_min[0]), 3), "max": round(float(pr_max[0]), 3)},
                                     "exchange_and_merge": {"min": round(float(pr_min[1]), 3), "max": round(float(pr_max[1]), 3)},
                                     "note": "host wall time per rank; a rank that finishes its shard early waits in the exchange for the slowest: min(exchange) is the step's own cost"},
            "exchange_ms_per_step": round(float(pr_min[1]), 3), "exchange_ldir": exchange_ldir,
            "power": {"timed_schedule": power_timed, "kernels_back_to_back": power_alone,
                      "what": "board power of rank 0's GPU sampled by a host thread during the timed steps, and during the two back-to-back steps after them (bound_cus 0); joules_per_query = mean watts x wall seconds / queries; null = the box exposes no power reading"},
            "ranks": ranks_info,
            "distinct_devices": len({(r_["uuid"] or r_["pci_bus_id"]) for r_ in ranks_info}),
            "shared_gpu": bool(a.share_gpu) or len({(r_["uuid"] or r_["pci_bus_id"]) for r_ in ranks_info}) < world,
            "rccl_ranks": ((xch.comm_count if xch.is_rccl else 0) if xch is not None else (dist.get_world_size() if (use_dist and a.backend == "nccl") else 0)),
            "rccl_ranks_is": ("ncclCommCount of the C++ exchange's communicator (csrc/rank_exchange.cpp)" if (xch is not None and xch.is_rccl) else
                              "0: the TCP stand-in carries the exchange, there is no RCCL communicator" if xch is not None else
                              "world size of torch.distributed's nccl (= RCCL) process group" if (use_dist and a.backend == "nccl") else "0: no RCCL communicator in this run (one rank, or a gloo test run)"),
            "scaling_curve_note": "no 1 -> 8 GPU curve has been measured by the builder: multi-GPU boxes are only available to the driver",
            "latency_ms_per_query": round(ms_per_step / max(1, Q), 3), "launch_groups_per_step": tm_acc["launch_groups"] // max(1, a.steps),
            "rank1_hits": f"{hits}/{Q}", "setup_s": {"generate": round(t_gen, 1), "upload": round(t_up, 1)},
        }
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(cb_bytes, lats, gal, lo)
            try:
                cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                cpu_model = "unknown"
            phys, logical = physical_cores()
            pps = cpu["pairs_per_s"]
            out["cpu_baseline"] = {"value": round(pps / G, 6), "unit": "queries/s", "cores": cpu["threads"], "threads": cpu["threads"],
                                   "usable_cpus": cpu["usable_cpus"], "physical_cores": phys, "logical_cpus": logical, "limits": cpu["limits"],
                                   "kind": "port", "sample": cpu["sample"], "cpu_model": cpu_model, "pairs_per_s": round(pps, 1),
                                   "thread_curve_pairs_per_s": cpu["curve"],
                                   "compute_only_8_threads_static16_queries_per_s": round(cpu["pairs_per_s_8_threads"] / G, 6),
                                   "reference_faithful_8_threads_static16_reparse_per_pair_queries_per_s": round(cpu["pairs_per_s_reference_faithful"] / G, 6),
                                   "sample_8_threads": cpu["sample_8_threads"],
                                   "note": "value = the best point of the compute-only thread curve (gallery resident in RAM, one pair at a time per thread); cores = the threads of that point; "
                                           "reference-faithful = the reference's own loop: 8 threads, schedule(static,16), every rolled .dat re-parsed per pair from page-cache-warm files (matcher.cpp:168-173)"}
            out["speedup_vs_cpu_baseline"] = round(value / (pps / G), 1)
    m.free_queries(qh)
    m.close()
    if xch is not None: xch.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)                              # the last (and only) line on stdout


if __name__ == "__main__":
    main()
