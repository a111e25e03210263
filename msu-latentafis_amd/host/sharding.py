"""Gallery sharding and rank-list merge.  Pure host logic: numpy for the arithmetic, torch.distributed only for
the one exchange step (all_gather of the per-shard top-k; backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).

Gallery templates are independent units: partition them into one contiguous shard per rank, balanced by the cost driver
(number of rolled texture points); every rank scores all queries against its shard and reports its local top-k with GLOBAL
gallery indices; the merged list is the top-k of the union, score descending, ties by ascending global index.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_bounds(cost: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) per rank with ~equal total cost.  Every rank gets a (possibly empty) range; ranges tile [0, G)."""
    G = len(cost)
    if world <= 1:
        return [(0, G)]
    c = np.concatenate([[0], np.cumsum(np.asarray(cost, dtype=np.float64))])
    total = c[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(c, target, side="left"))
        k = min(max(k, cuts[-1]), G)
        cuts.append(k)
    cuts.append(G)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def merge_topk(idx: np.ndarray, score: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """idx, score: [R, Q, kk] per-shard rank lists (idx -1 = padding).  Returns the merged [Q, k] list."""
    R, Q, kk = idx.shape
    fi = np.transpose(idx, (1, 0, 2)).reshape(Q, R * kk)
    fs = np.transpose(score, (1, 0, 2)).reshape(Q, R * kk).astype(np.float32)
    out_i = np.full((Q, k), -1, np.int64); out_s = np.full((Q, k), -np.inf, np.float32)
    for q in range(Q):
        valid = fi[q] >= 0
        vi, vs = fi[q][valid], fs[q][valid]
        order = np.lexsort((vi, -vs.astype(np.float64)))[:k]      # score descending, then global index ascending
        out_i[q, :len(order)] = vi[order]; out_s[q, :len(order)] = vs[order]
    return out_i, out_s


def gather_topk(idx: np.ndarray, score: np.ndarray, k: int, device=None, force: bool = False):
    """The one exchange step: all_gather of [Q, kk] (int64 idx, f32 score) from every rank, then merge on every rank.
    Messages are tiny (24 x 12 B per query per rank); this is latency-, not bandwidth-bound."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return merge_topk(idx[None], score[None], k)
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    ti = torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
    ts = torch.from_numpy(np.ascontiguousarray(score)).to(dev)
    gi = [torch.empty_like(ti) for _ in range(world)]
    gs = [torch.empty_like(ts) for _ in range(world)]
    dist.all_gather(gi, ti)
    dist.all_gather(gs, ts)
    ai = torch.stack(gi).cpu().numpy(); as_ = torch.stack(gs).cpu().numpy()
    return merge_topk(ai, as_, k)


class CppExchange:
    """The exchange step of the C++ `match` host (csrc/rank_exchange.cpp: ncclAllGather over RCCL / xGMI on a dedicated HIP stream, staged
    through device buffers; AFIS_EXCHANGE=tcp swaps in the TCP stand-in that lets ranks share a GPU), bound with ctypes from
    libafis_exchange.so.  Ranks and rendezvous come from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (the id exchange uses
    MASTER_PORT + 1, IPv4 literal only).  No fallback: a missing library raises."""

    def __init__(self, device: int):
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "libafis_exchange.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `make -C msu-latentafis_amd/csrc`")
        self._C = C
        lib = self.lib = C.CDLL(path)
        lib.afis_exchange_create.restype = C.c_void_p; lib.afis_exchange_create.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        lib.afis_exchange_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.afis_exchange_last_error.restype = C.c_char_p; lib.afis_exchange_last_error.argtypes = [C.c_void_p]
        lib.afis_exchange_destroy.restype = None; lib.afis_exchange_destroy.argtypes = [C.c_void_p]
        for f in ("afis_exchange_world", "afis_exchange_rank", "afis_exchange_is_rccl", "afis_exchange_comm_count", "afis_exchange_comm_device"):
            getattr(lib, f).argtypes = [C.c_void_p]
        err = C.create_string_buffer(512)
        self.h = lib.afis_exchange_create(device, err, 512)
        if not self.h:
            raise RuntimeError("afis_exchange_create: " + err.value.decode(errors="replace"))
        self.world = lib.afis_exchange_world(self.h); self.rank = lib.afis_exchange_rank(self.h)
        self.is_rccl = bool(lib.afis_exchange_is_rccl(self.h))
        self.comm_count = int(lib.afis_exchange_comm_count(self.h))         # ncclCommCount of this rank's communicator (-1: the TCP stand-in has none)
        self.comm_device = int(lib.afis_exchange_comm_device(self.h))       # ncclCommCuDevice

    def all_gather(self, block: np.ndarray) -> np.ndarray:
        """block: any contiguous array, the same shape and dtype on every rank -> [world, *block.shape]."""
        b = np.ascontiguousarray(block)
        out = np.empty((self.world,) + b.shape, b.dtype)
        rc = self.lib.afis_exchange_all_gather(self.h, b.ctypes.data, out.ctypes.data, b.nbytes)
        if rc != 0:
            raise RuntimeError("afis_exchange_all_gather: " + self.lib.afis_exchange_last_error(self.h).decode(errors="replace"))
        return out

    def gather_topk(self, idx: np.ndarray, score: np.ndarray, k: int):
        """One all-gather of the per-rank block [Q][kk] x (idx i64, score f32), as `match -l` sends it, then the merge."""
        Q, kk = idx.shape
        blk = np.empty(Q * kk * 12, np.uint8)
        blk[:Q * kk * 8] = np.ascontiguousarray(idx, np.int64).view(np.uint8).ravel()
        blk[Q * kk * 8:] = np.ascontiguousarray(score, np.float32).view(np.uint8).ravel()
        allb = self.all_gather(blk)
        ai = allb[:, :Q * kk * 8].copy().view(np.int64).reshape(self.world, Q, kk)
        as_ = allb[:, Q * kk * 8:].copy().view(np.float32).reshape(self.world, Q, kk)
        return merge_topk(ai, as_, k)

    def close(self):
        if self.h:
            self.lib.afis_exchange_destroy(self.h); self.h = None
