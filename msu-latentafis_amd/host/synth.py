"""Seeded synthetic latents and rolled galleries with PLANTED MATES.

A random (non-mate) rolled template scores exactly 0 against any latent because the geometric-consistency
filters reject every correspondence, so a gallery without planted mates has a degenerate 100k-way tie.  Each
latent therefore gets one true mate and a few partial mates: a translated copy of a subset of its minutiae and
texture points with descriptor noise, texture codes = PQ-encode(latent texture descriptors + noise).

Two products:
  * FPTemplate objects / .dat bytes for small cases (tests, CLI);
  * a PackedGallery (concatenated SoA arrays + CSR offsets) for large galleries, which is what
    afis_gallery_add_packed() takes without a per-template Python loop.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .templates import (DESCRIPTOR_NORM, Codebook, FPTemplate, MinutiaeTemplate, TextureTemplate)

IMG_W, IMG_H = 768, 800
BLK_W, BLK_H = 45, 47        # texture block grid ((px-24)/16)


def _unit(d: np.ndarray) -> np.ndarray:
    d = np.asarray(d, dtype=np.float32)
    n2 = np.einsum("...i,...i->...", d, d)
    return (d * (np.float32(DESCRIPTOR_NORM) / np.sqrt(n2))[..., None]).astype(np.float32)


def make_latent(rng: np.random.Generator, n_minu_tpl: int = 28, n_tex_lo: int = 400, n_tex_hi: int = 1000,
                n_minu_lo: int = 20, n_minu_hi: int = 60) -> FPTemplate:
    """28 minutiae templates (7 minutiae sets x 4 images in the reference, extraction_latent.py:117-181) that
    are noisy views of one base pool, plus one texture template with two orientations per grid point
    (extraction_latent.py:187-212)."""
    pool = n_minu_hi
    bx = rng.integers(60, IMG_W - 60, pool); by = rng.integers(60, IMG_H - 60, pool)
    bo = rng.uniform(-np.pi, np.pi, pool).astype(np.float32)
    bd = _unit(rng.standard_normal((pool, 96)))
    t = FPTemplate()
    for _ in range(n_minu_tpl):
        n = int(rng.integers(n_minu_lo, n_minu_hi + 1))
        sel = np.sort(rng.permutation(pool)[:n])
        des = _unit(bd[sel] + rng.standard_normal((n, 96)) * 0.02)
        t.minu.append(MinutiaeTemplate(bx[sel].astype(np.int16), by[sel].astype(np.int16), bo[sel].copy(), des))
    n_tex = int(rng.integers(n_tex_lo, n_tex_hi + 1))
    n_grid = max(1, n_tex // 2)
    cells = rng.permutation(BLK_W * BLK_H)[:n_grid]
    gx = (cells % BLK_W).astype(np.int16); gy = (cells // BLK_W).astype(np.int16)
    go = rng.uniform(-np.pi / 2, np.pi / 2, n_grid).astype(np.float32)
    tx = np.concatenate([gx, gx]); ty = np.concatenate([gy, gy])
    to = np.concatenate([go, go + np.float32(np.pi)]).astype(np.float32)
    td = _unit(rng.standard_normal((2 * n_grid, 96)))
    t.tex.append(TextureTemplate(tx, ty, to, des=td))
    t._pool = (bx, by, bo, bd)    # generator-private: lets make_mate plant consistent minutiae
    return t


def make_rolled(rng: np.random.Generator, cb: Codebook, n_minu: Optional[int] = None, n_tex: Optional[int] = None) -> FPTemplate:
    """Random non-mate rolled template: 1 minutiae template + 1 PQ texture template (extraction_rolled.py:105-141)."""
    if n_minu is None:
        n_minu = int(np.clip(round(rng.normal(80, 15)), 20, 200))
    if n_tex is None:
        n_tex = int(rng.integers(600, 1001))
    t = FPTemplate()
    t.minu.append(MinutiaeTemplate(rng.integers(0, IMG_W, n_minu).astype(np.int16), rng.integers(0, IMG_H, n_minu).astype(np.int16),
                                   rng.uniform(-np.pi, np.pi, n_minu).astype(np.float32), _unit(rng.standard_normal((n_minu, 96)))))
    t.tex.append(TextureTemplate(rng.integers(0, BLK_W, n_tex).astype(np.int16), rng.integers(0, BLK_H, n_tex).astype(np.int16),
                                 rng.uniform(-np.pi / 2, np.pi / 2, n_tex).astype(np.float32),
                                 codes=rng.integers(0, cb.K, (n_tex, cb.M)).astype(np.uint8)))
    return t


def make_mate(rng: np.random.Generator, cb: Codebook, latent: FPTemplate, frac: float = 0.75, sigma: float = 0.08,
              n_minu: Optional[int] = None, n_tex: Optional[int] = None) -> FPTemplate:
    """Rolled template that shares `frac` of the latent's minutiae pool and texture grid, translated rigidly
    (pixels for minutiae, whole blocks for texture) with descriptor noise `sigma`."""
    t = make_rolled(rng, cb, n_minu, n_tex)
    bx, by, bo, bd = latent._pool
    m = t.minu[0]
    k = min(int(len(bx) * frac), m.n)
    sel = rng.permutation(len(bx))[:k]
    dx, dy = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
    m.x[:k] = np.clip(bx[sel] + dx, 0, IMG_W - 1); m.y[:k] = np.clip(by[sel] + dy, 0, IMG_H - 1)
    m.ori[:k] = bo[sel]
    m.des[:k] = _unit(bd[sel] + rng.standard_normal((k, 96)) * sigma)
    lt = latent.tex[0]; rt = t.tex[0]
    half = lt.n // 2                                        # one orientation per grid point in a rolled print
    kt = min(int(half * frac), rt.n)
    selt = rng.permutation(half)[:kt]
    bdx, bdy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
    rt.x[:kt] = np.clip(lt.x[selt] + bdx, 0, BLK_W - 1); rt.y[:kt] = np.clip(lt.y[selt] + bdy, 0, BLK_H - 1)
    rt.ori[:kt] = lt.ori[selt]
    rt.codes[:kt] = cb.encode(lt.des[selt] + rng.standard_normal((kt, 96)).astype(np.float32) * (sigma * 0.5))
    return t


@dataclass
class PackedGallery:
    """Concatenated SoA gallery (one minutiae + one texture template per entry), CSR offsets."""
    minu_off: np.ndarray   # int64 [G+1]
    minu_x: np.ndarray     # int16 [sum n]
    minu_y: np.ndarray
    minu_ori: np.ndarray   # float32
    minu_des: np.ndarray   # float32 [sum n, 96]
    tex_off: np.ndarray    # int64 [G+1]
    tex_x: np.ndarray      # int16
    tex_y: np.ndarray
    tex_ori: np.ndarray
    tex_codes: np.ndarray  # uint8 [sum n, 16]

    @property
    def G(self) -> int: return len(self.minu_off) - 1

    def template(self, g: int) -> FPTemplate:
        a, b = int(self.minu_off[g]), int(self.minu_off[g + 1])
        c, d = int(self.tex_off[g]), int(self.tex_off[g + 1])
        t = FPTemplate()
        if b > a:
            t.minu.append(MinutiaeTemplate(self.minu_x[a:b].copy(), self.minu_y[a:b].copy(), self.minu_ori[a:b].copy(), self.minu_des[a:b].copy()))
        if d > c:
            t.tex.append(TextureTemplate(self.tex_x[c:d].copy(), self.tex_y[c:d].copy(), self.tex_ori[c:d].copy(), codes=self.tex_codes[c:d].copy()))
        return t

    def set_template(self, g: int, t: FPTemplate) -> None:
        """Overwrite entry g in place (counts must match) — used to plant mates."""
        a, b = int(self.minu_off[g]), int(self.minu_off[g + 1])
        c, d = int(self.tex_off[g]), int(self.tex_off[g + 1])
        m, x = t.minu[0], t.tex[0]
        assert m.n == b - a and x.n == d - c
        self.minu_x[a:b] = m.x; self.minu_y[a:b] = m.y; self.minu_ori[a:b] = m.ori; self.minu_des[a:b] = m.des
        self.tex_x[c:d] = x.x; self.tex_y[c:d] = x.y; self.tex_ori[c:d] = x.ori; self.tex_codes[c:d] = x.codes

    def slice(self, lo: int, hi: int) -> "PackedGallery":
        """Contiguous shard [lo, hi) with offsets rebased (gallery sharding)."""
        a, b = int(self.minu_off[lo]), int(self.minu_off[hi])
        c, d = int(self.tex_off[lo]), int(self.tex_off[hi])
        return PackedGallery(self.minu_off[lo:hi + 1] - a, self.minu_x[a:b], self.minu_y[a:b], self.minu_ori[a:b], self.minu_des[a:b],
                             self.tex_off[lo:hi + 1] - c, self.tex_x[c:d], self.tex_y[c:d], self.tex_ori[c:d], self.tex_codes[c:d])


GEN_BLOCK = 1024   # templates per independently seeded block: any shard can be generated without the rest

# Named workloads of bench.py / the sweeps.  "headline" = SURVEY section 8d's shapes (BASELINE.json's ~80 minutiae per rolled print); "wide" = the shapes the reference's
# reader also accepts but the synthetic envelope never produced (matcher.cpp:788-790 allows 2000 minutiae per template, extraction_rolled.py:105-108 caps nothing for rolled
# prints): rolled minutiae clip(N(130, 40), 20, 400), latent minutiae U{20..150}; texture templates as in the headline.
WORKLOADS = {
    "headline": {"gallery": {}, "latent": {}},
    "wide": {"gallery": {"n_minu_mean": 130, "n_minu_sd": 40, "n_minu_max": 400}, "latent": {"n_minu_lo": 20, "n_minu_hi": 150}},
    # the headline's template sizes with the STRUCTURE of extracted prints (host/synth_structured.py): grid coordinates inside a foreground blob, smooth ridge flow,
    # descriptors near a shared manifold that are PQ-encoded afterwards (neighbouring points share codes; a chosen share of exact duplicates)
    "structured": {"gallery": {}, "latent": {}},
}


def gallery_counts(seed: int, G: int, n_minu_mean: float = 80, n_tex_lo: int = 600, n_tex_hi: int = 1000, n_minu_sd: float = 15, n_minu_max: int = 200):
    """Per-template point counts of the synthetic gallery (cheap; every rank computes all of them to place shard bounds)."""
    rng = np.random.default_rng([seed, 0xC0])
    nm = np.clip(np.rint(rng.normal(n_minu_mean, n_minu_sd, G)), 20, n_minu_max).astype(np.int64)
    nt = rng.integers(n_tex_lo, n_tex_hi + 1, G).astype(np.int64)
    return nm, nt


def make_packed_gallery(seed: int, G: int, cb: Codebook, lo: int = 0, hi: Optional[int] = None, n_minu_mean: float = 80,
                        n_tex_lo: int = 600, n_tex_hi: int = 1000, n_minu_sd: float = 15, n_minu_max: int = 200) -> PackedGallery:
    """Vectorised random (non-mate) gallery, templates [lo, hi) of a G-template gallery; plant mates afterwards with
    plant_mates().  Content depends only on (seed, G, template index), not on the shard bounds."""
    hi = G if hi is None else hi
    nm_all, nt_all = gallery_counts(seed, G, n_minu_mean, n_tex_lo, n_tex_hi, n_minu_sd, n_minu_max)
    nm, nt = nm_all[lo:hi], nt_all[lo:hi]
    mo = np.concatenate([[0], np.cumsum(nm)]); to = np.concatenate([[0], np.cumsum(nt)])
    NM, NT = int(mo[-1]), int(to[-1])
    mx = np.empty(NM, np.int16); my = np.empty(NM, np.int16); mori = np.empty(NM, np.float32); des = np.empty((NM, 96), np.float32)
    tx = np.empty(NT, np.int16); ty = np.empty(NT, np.int16); tori = np.empty(NT, np.float32); codes = np.empty((NT, cb.M), np.uint8)
    for b in range(lo // GEN_BLOCK, (max(hi, 1) - 1) // GEN_BLOCK + 1):
        b_lo, b_hi = b * GEN_BLOCK, min(G, (b + 1) * GEN_BLOCK)
        rng = np.random.default_rng([seed, 1, b])
        bm, bt = int(nm_all[b_lo:b_hi].sum()), int(nt_all[b_lo:b_hi].sum())
        bmx = rng.integers(0, IMG_W, bm).astype(np.int16); bmy = rng.integers(0, IMG_H, bm).astype(np.int16)
        bmo = rng.uniform(-np.pi, np.pi, bm).astype(np.float32)
        bd = _unit(rng.random((bm, 96), dtype=np.float32) - np.float32(0.5))   # uniform cube directions: 5x cheaper than Gaussians at 10^7 rows
        btx = rng.integers(0, BLK_W, bt).astype(np.int16); bty = rng.integers(0, BLK_H, bt).astype(np.int16)
        bto = rng.uniform(-np.pi / 2, np.pi / 2, bt).astype(np.float32)
        bc = (np.frombuffer(rng.bytes(bt * cb.M), np.uint8).reshape(bt, cb.M) if cb.K == 256 else rng.integers(0, cb.K, (bt, cb.M)).astype(np.uint8))
        # intersect the block with [lo, hi)
        s_lo, s_hi = max(lo, b_lo), min(hi, b_hi)
        if s_hi <= s_lo:
            continue
        m_skip = int(nm_all[b_lo:s_lo].sum()); m_take = int(nm_all[s_lo:s_hi].sum())
        t_skip = int(nt_all[b_lo:s_lo].sum()); t_take = int(nt_all[s_lo:s_hi].sum())
        md = int(mo[s_lo - lo]); td = int(to[s_lo - lo])
        mx[md:md + m_take] = bmx[m_skip:m_skip + m_take]; my[md:md + m_take] = bmy[m_skip:m_skip + m_take]
        mori[md:md + m_take] = bmo[m_skip:m_skip + m_take]; des[md:md + m_take] = bd[m_skip:m_skip + m_take]
        tx[td:td + t_take] = btx[t_skip:t_skip + t_take]; ty[td:td + t_take] = bty[t_skip:t_skip + t_take]
        tori[td:td + t_take] = bto[t_skip:t_skip + t_take]; codes[td:td + t_take] = bc[t_skip:t_skip + t_take]
    return PackedGallery(mo, mx, my, mori, des, to, tx, ty, tori, codes)


def mate_slots(seed: int, G: int, n_latents: int, n_partial: int = 3):
    """Global gallery indices that carry the planted mates of each latent: [n_latents, 1 + n_partial], no collisions."""
    rng = np.random.default_rng([seed, 0xA7])
    per = 1 + n_partial
    if n_latents * per > G:
        per = max(1, G // max(1, n_latents))
    return rng.permutation(G)[:n_latents * per].reshape(n_latents, per)


MATE_FRACS = [0.8, 0.5, 0.35, 0.25, 0.2, 0.15]


def plant_mates(seed: int, gal: PackedGallery, cb: Codebook, latents: List[FPTemplate], G: Optional[int] = None, lo: int = 0,
                n_partial: int = 3) -> Dict[int, List[Tuple[int, float]]]:
    """For each latent overwrite 1 + n_partial gallery entries with mates of decreasing overlap.  `gal` holds templates
    [lo, lo + gal.G) of a G-template gallery; only the slots inside that range are written, and a mate's content depends only on
    (seed, latent index, slot rank), so every shard plants the same mates.  Returns {latent: [(GLOBAL gallery index, frac), ...]}."""
    G = gal.G if G is None else G
    slots = mate_slots(seed, G, len(latents), n_partial)
    planted: Dict[int, List[Tuple[int, float]]] = {}
    for q, L in enumerate(latents):
        planted[q] = []
        for r, s_ in enumerate(slots[q]):
            g = int(s_); frac = MATE_FRACS[min(r, len(MATE_FRACS) - 1)]
            planted[q].append((g, frac))
            if not (lo <= g < lo + gal.G):
                continue
            k = g - lo
            nm = int(gal.minu_off[k + 1] - gal.minu_off[k]); nt = int(gal.tex_off[k + 1] - gal.tex_off[k])
            rng = np.random.default_rng([seed, 2, q, r])
            gal.set_template(k, make_mate(rng, cb, L, frac=frac, sigma=0.08, n_minu=nm, n_tex=nt))
    return planted


def make_latents(seed: int, n: int, **kw) -> List[FPTemplate]:
    return [make_latent(np.random.default_rng([seed, 3, i]), **kw) for i in range(n)]


# ---- off-envelope shapes (tools/offenv_sweep.py, tests/test_gpu_parity.py) --------------------------------------------------------
# What the reference's reader accepts and the synthetic envelope of SURVEY section 8d never produces: up to 2000 minutiae per template
# (matcher.cpp:788-790), rolled prints of any size (extraction_rolled.py:105-108 caps nothing), texture templates beyond the scorer's 1000-row clamp
# (matcher.cpp:544-547), pixel coordinates on both sides of 2047 (where the graph kernels' packed 16-bit arithmetic ends).
OFFENV_ROLLED_MINUTIAE = [129, 160, 200, 257, 400, 1000, 2000]
OFFENV_LATENT_MINUTIAE = [65, 100, 128, 200]


def _shift_minutiae(t: FPTemplate, dx: int, dy: int) -> None:
    for m in t.minu:
        m.x[:] = (m.x.astype(np.int32) + dx).astype(np.int16); m.y[:] = (m.y.astype(np.int32) + dy).astype(np.int16)


def make_offenvelope_set(seed: int, n_latents: int, n_rolled: int, cb: Codebook, mates_per_latent: int = 2):
    """n_latents latents and n_rolled rolled templates off the synthetic envelope: latent minutiae templates of OFFENV_LATENT_MINUTIAE (+-1) minutiae and
    1001..1900 (or 400..1000) texture rows, rolled minutiae templates of OFFENV_ROLLED_MINUTIAE (+ 0..2) minutiae and 600..1900 texture points; every latent has
    `mates_per_latent` planted mates among the rolled templates (slots q * mates .. ); half of the templates are translated so that their pixel coordinates straddle 2047.
    Returns (latents, rolled, {latent: [rolled indices of its mates]})."""
    rng = np.random.default_rng([seed, 0x0FE])
    lats: List[FPTemplate] = []
    shifts = []
    for q in range(n_latents):
        nl = int(rng.choice(OFFENV_LATENT_MINUTIAE)) + int(rng.integers(-1, 2))
        if rng.random() < 0.6: tl, th = 1001, 1900
        else: tl, th = 400, 1000
        L = make_latent(rng, n_tex_lo=tl, n_tex_hi=th, n_minu_lo=max(2, nl - int(rng.integers(0, 3))), n_minu_hi=nl)
        lats.append(L)
        shifts.append((int(rng.choice([0, 1300, 1500])), int(rng.choice([0, 1280]))))
    rolled: List[FPTemplate] = []
    mates: Dict[int, List[int]] = {q: [] for q in range(n_latents)}
    for g in range(n_rolled):
        nm = int(rng.choice(OFFENV_ROLLED_MINUTIAE)) + int(rng.integers(0, 3))
        nm = min(nm, 2000)
        nt = int(rng.choice([600, 800, 1000, 1001, 1500, 1900]))
        q = g // mates_per_latent
        if q < n_latents:
            R = make_mate(rng, cb, lats[q], frac=float(rng.uniform(0.3, 0.9)), n_minu=nm, n_tex=nt)
            dx, dy = shifts[q]
            dx += int(rng.choice([0, 0, 600])); dy += int(rng.choice([0, 0, 700]))      # a mate may sit on the other side of 2047 than its latent: translation-invariant stages must not care
            _shift_minutiae(R, dx, dy)
            mates[q].append(g)
        else:
            R = make_rolled(rng, cb, n_minu=nm, n_tex=nt)
            _shift_minutiae(R, int(rng.choice([0, 0, 1300, 1500])), int(rng.choice([0, 1280])))
        rolled.append(R)
    for q, L in enumerate(lats):
        _shift_minutiae(L, *shifts[q])
        if hasattr(L, "_pool"): del L._pool
    return lats, rolled, mates
