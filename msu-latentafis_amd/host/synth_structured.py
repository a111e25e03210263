"""Seeded synthetic templates with the STRUCTURE of extracted fingerprints (bench.py --workload structured, tools/parity_sweep.py --structured).

host/synth.py draws every texture point's coordinates, orientation and PQ code bytes independently (SURVEY section 8d's generator).  The reference's
extractor produces nothing like that:

  * rolled texture template (extraction/extraction_rolled.py:112-141): virtual minutiae on a regular 16-px grid, visited y-major / x-minor, only where the
    foreground mask is more than 24 px thick, at most the first 1000 in scan order; orientation = minus the ridge-flow direction of the point's block
    (a smooth field); descriptors from overlapping 96-px patches, THEN PQ-encoded: neighbouring points share most or all of their 16 codes.
  * latent texture template (extraction/extraction_latent.py:187-212): the same grid, two virtual minutiae per point (ori, pi + ori), fp32 descriptors.
  * minutiae lie in the foreground and point along the ridge flow (either sense).

This generator reproduces that structure:

  * foreground = the n cells of the 45 x 47 block grid that are nearest to a random centre under a randomly rotated, low-frequency-perturbed elliptical norm:
    unique coordinates, one connected blob, in scan order (n is drawn first, so template sizes follow the same distribution as synth.py's);
  * ridge flow = a zero-pole model (Sherlock & Monro): arch (no singular point), loop (core + delta) or whorl (two of each), orientation mod pi;
  * descriptors = unit-norm (1.73) points near a low-dimensional manifold that all prints share: des = W z(x, y) + sigma * noise with W a fixed 96 x D matrix
    (the "network") and z a smooth per-template field (cosine basis of order <= 2 per axis over the grid).  sigma sets how many of a template's points carry a
    code vector that another point of the same template also carries (`dup_share()` measures it; the named levels of DUP_SIGMA give about 0 / 10 / 30 %);
  * mates: a rolled template whose field, on the cells that are a translated copy of a subset of the latent's cells, is the latent's (orientation copied, codes =
    encode(latent descriptor + noise)); its own cells everywhere else.  Coordinates stay unique.

Everything depends only on (seed, template index), as in synth.py; a block of templates is generated with a few vectorised numpy calls.  `encode` lets the caller
supply the nearest-codeword encoder (the GPU's afis_pq_encode for 10^8 points; Codebook.encode_fast on the CPU).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from .synth import BLK_H, BLK_W, GEN_BLOCK, IMG_H, IMG_W, MATE_FRACS, PackedGallery, _unit, gallery_counts, mate_slots
from .templates import DESCRIPTOR_NORM, Codebook, FPTemplate, MinutiaeTemplate, TextureTemplate

# Weight of the field's constant term — what all descriptors of ONE print have in common — against 1/2, 1/3 ... for the orders of its variation over the print.  At 1.0 two prints
# whose constant terms point away from each other have almost only NEGATIVE descriptor products: 8 % of the (latent, rolled) minutiae pairs then keep fewer than 120 positive
# similarities after the clamp of matcher.cpp:447-451, and their candidate lists are filled up with zero entries (whose order is the sort's tie order: tools/tie_site_sweep.py).
# A patch descriptor describes local ridge structure, not the finger: 0.3 (0.15 % such pairs) is the named workload; tests also run 1.0 for the sake of that path.
IDENTITY_WEIGHT = 0.3
MANIFOLD_DIMS = 12                       # D: dimensions of the descriptor manifold all templates share
N_BASIS = 9                              # cosine basis functions of the per-template field: orders (0..2) x (0..2)
# descriptor noise that leaves about this share of a rolled template's points with a code vector some other point of the template also has
# (calibrated at the headline's template sizes with the shipped codebook: tools/structured_calibrate.py)
DUP_SIGMA = {0: 0.030, 10: 0.0066, 30: 0.0017}


def _network() -> np.ndarray:
    """The fixed 96 x D matrix that maps a field value to a descriptor direction (one for all templates: a property of the extractor, not of a print)."""
    rng = np.random.default_rng(0x5EED0D)
    return rng.standard_normal((96, MANIFOLD_DIMS)).astype(np.float32)


_W = _network()


def _basis(cx: np.ndarray, cy: np.ndarray) -> np.ndarray:
    """[n, N_BASIS] cosine basis at block coordinates (cx, cy) (may be fractional: minutiae sit between grid points)."""
    u = np.asarray(cx, np.float32) * np.float32(np.pi / BLK_W); v = np.asarray(cy, np.float32) * np.float32(np.pi / BLK_H)
    cols = [np.cos(a * u) * np.cos(b * v) for a in range(3) for b in range(3)]
    return np.stack(cols, axis=-1).astype(np.float32)


def _field_coeffs(rng: np.random.Generator, n: int) -> np.ndarray:
    """[n, D, N_BASIS] coefficients of n templates' fields: the constant term carries a print's identity, the higher orders its variation over the print."""
    a = rng.standard_normal((n, MANIFOLD_DIMS, N_BASIS)).astype(np.float32)
    order = np.array([a_ + b_ for a_ in range(3) for b_ in range(3)], np.float32)
    w = np.float32(1.0) / (np.float32(1.0) + order)
    w[0] = np.float32(IDENTITY_WEIGHT)
    return a * w[None, None, :]


def _flow_params(rng: np.random.Generator, n: int) -> np.ndarray:
    """[n, 10]: theta0, class (0 arch / 1 loop / 2 whorl), core 1 (x, y), delta 1, core 2, delta 2 — in block units."""
    p = np.zeros((n, 10), np.float32)
    p[:, 0] = rng.uniform(-0.3, 0.3, n)
    p[:, 1] = rng.choice([0, 1, 2], n, p=[0.15, 0.6, 0.25])
    p[:, 2] = rng.uniform(0.35, 0.65, n) * BLK_W; p[:, 3] = rng.uniform(0.30, 0.55, n) * BLK_H          # core 1
    side = rng.choice([-1.0, 1.0], n)
    p[:, 4] = p[:, 2] + side * rng.uniform(0.2, 0.35, n) * BLK_W; p[:, 5] = rng.uniform(0.7, 0.95, n) * BLK_H   # delta 1 (below, to one side)
    p[:, 6] = p[:, 2] + rng.uniform(-0.08, 0.08, n) * BLK_W; p[:, 7] = p[:, 3] + rng.uniform(0.08, 0.2, n) * BLK_H   # core 2 (whorl)
    p[:, 8] = p[:, 2] - side * rng.uniform(0.2, 0.35, n) * BLK_W; p[:, 9] = rng.uniform(0.7, 0.95, n) * BLK_H   # delta 2
    return p


def flow_direction(p: np.ndarray, cx: np.ndarray, cy: np.ndarray) -> np.ndarray:
    """Ridge-flow direction in (-pi/2, pi/2] at block coordinates (cx, cy); p = one row of _flow_params per point (or one row for all)."""
    p = np.broadcast_to(p, (len(cx), 10)) if p.ndim == 1 else p
    cx = np.asarray(cx, np.float32) + np.float32(0.013); cy = np.asarray(cy, np.float32) + np.float32(0.017)      # never exactly on a singular point
    th = p[:, 0].copy()
    arch = np.float32(0.35) * np.sin((cx / BLK_W - np.float32(0.5)) * np.float32(np.pi)) * (cy / BLK_H)          # a gentle arch for every class
    th = th - arch
    k1 = p[:, 1] >= 1; k2 = p[:, 1] >= 2
    a = np.arctan2(cy - p[:, 3], cx - p[:, 2]) - np.arctan2(cy - p[:, 5], cx - p[:, 4])
    b = np.arctan2(cy - p[:, 7], cx - p[:, 6]) - np.arctan2(cy - p[:, 9], cx - p[:, 8])
    th = th + np.float32(0.5) * (np.where(k1, a, 0) + np.where(k2, b, 0))
    th = np.mod(th + np.float32(np.pi / 2), np.float32(np.pi)) - np.float32(np.pi / 2)
    return th.astype(np.float32)


def _blob_params(rng: np.random.Generator, n: int) -> np.ndarray:
    """[n, 8]: centre (x, y), axis ratio, rotation, two perturbation amplitudes and phases."""
    p = np.zeros((n, 8), np.float32)
    p[:, 0] = rng.uniform(0.4, 0.6, n) * BLK_W; p[:, 1] = rng.uniform(0.4, 0.6, n) * BLK_H
    p[:, 2] = rng.uniform(0.75, 1.3, n); p[:, 3] = rng.uniform(-0.4, 0.4, n)
    p[:, 4] = rng.uniform(0.0, 0.12, n); p[:, 5] = rng.uniform(0.0, 0.08, n)
    p[:, 6] = rng.uniform(0, 2 * np.pi, n); p[:, 7] = rng.uniform(0, 2 * np.pi, n)
    return p


_CELL_X = (np.arange(BLK_W * BLK_H) % BLK_W).astype(np.float32)
_CELL_Y = (np.arange(BLK_W * BLK_H) // BLK_W).astype(np.float32)


def _blob_masks(bp: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """bool [n, BLK_H * BLK_W]: template t's foreground = its counts[t] nearest cells under its blob norm (cell index = y * BLK_W + x: scan order)."""
    dx = _CELL_X[None, :] - bp[:, 0:1]; dy = _CELL_Y[None, :] - bp[:, 1:2]
    c, s = np.cos(bp[:, 3:4]), np.sin(bp[:, 3:4])
    u = (c * dx + s * dy) * bp[:, 2:3]; v = (-s * dx + c * dy) / bp[:, 2:3]
    ang = np.arctan2(v, u)
    r = np.sqrt(u * u + v * v) * (1 + bp[:, 4:5] * np.cos(2 * ang + bp[:, 6:7]) + bp[:, 5:6] * np.cos(3 * ang + bp[:, 7:8]))
    rank = np.argsort(np.argsort(r, axis=1, kind="stable"), axis=1, kind="stable")
    return rank < np.asarray(counts)[:, None]


def _descriptors(coef: np.ndarray, tidx: np.ndarray, cx: np.ndarray, cy: np.ndarray, sigma: float, rng: np.random.Generator, device=None) -> np.ndarray:
    """unit(W z + sigma * noise) for points (template tidx[i], block coordinates cx[i], cy[i]); coef = [n_templates, D, N_BASIS].  device: a torch device that does the
    arithmetic and draws the noise (its own seeded generator: the values differ from the numpy path's, the distribution does not) — 10^8 points take minutes in numpy."""
    B = _basis(cx, cy)                                                      # [n, NB]
    if device is not None:
        import torch
        gen = torch.Generator(device=device); gen.manual_seed(int(rng.integers(0, 2 ** 62)))
        z = torch.einsum("ndb,nb->nd", torch.from_numpy(coef).to(device)[torch.from_numpy(np.asarray(tidx, np.int64)).to(device)], torch.from_numpy(B).to(device))
        d = z @ torch.from_numpy(_W).to(device).T
        d = d / d.norm(dim=1, keepdim=True)
        if sigma > 0:
            d = d + (torch.rand(d.shape, generator=gen, device=device, dtype=torch.float32) - 0.5) * float(sigma * np.sqrt(12.0))
        d = d * (float(DESCRIPTOR_NORM) / d.norm(dim=1, keepdim=True))
        return d.cpu().numpy()
    z = np.einsum("ndb,nb->nd", coef[tidx], B, optimize=True)              # [n, D]
    d = z @ _W.T                                                            # [n, 96]
    d *= (np.float32(1.0) / np.sqrt(np.einsum("ij,ij->i", d, d)))[:, None]
    if sigma > 0:
        d += (rng.random(d.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(sigma * np.sqrt(12.0))   # uniform noise, per-component sd = sigma (the direction has unit length: sd 0.102 per component)
    return _unit(d)


def _minutiae_in_blob(rng: np.random.Generator, mask_row_cells: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
    cells = mask_row_cells[rng.integers(0, len(mask_row_cells), n)]
    px = 24 + 16 * (cells % BLK_W) + rng.integers(-8, 8, n); py = 24 + 16 * (cells // BLK_W) + rng.integers(-8, 8, n)
    return np.clip(px, 0, IMG_W - 1).astype(np.int16), np.clip(py, 0, IMG_H - 1).astype(np.int16)


MINU_SIGMA = 0.10            # minutiae descriptors: the field at the minutia + this much own component per dimension


def make_packed_gallery_structured(seed: int, G: int, cb: Codebook, lo: int = 0, hi: Optional[int] = None, sigma: float = DUP_SIGMA[10],
                                   encode: Optional[Callable[[np.ndarray], np.ndarray]] = None, device=None, **count_kw) -> PackedGallery:
    """Templates [lo, hi) of a structured G-template gallery (same counts as synth.make_packed_gallery with the same seed and count arguments)."""
    hi = G if hi is None else hi
    encode = encode or cb.encode_fast
    nm_all, nt_all = gallery_counts(seed, G, **count_kw)
    nm, nt = nm_all[lo:hi], nt_all[lo:hi]
    mo = np.concatenate([[0], np.cumsum(nm)]); to = np.concatenate([[0], np.cumsum(nt)])
    NM, NT = int(mo[-1]), int(to[-1])
    mx = np.empty(NM, np.int16); my = np.empty(NM, np.int16); mori = np.empty(NM, np.float32); des = np.empty((NM, 96), np.float32)
    tx = np.empty(NT, np.int16); ty = np.empty(NT, np.int16); tori = np.empty(NT, np.float32); codes = np.empty((NT, cb.M), np.uint8)
    for b in range(lo // GEN_BLOCK, (max(hi, 1) - 1) // GEN_BLOCK + 1):
        b_lo, b_hi = b * GEN_BLOCK, min(G, (b + 1) * GEN_BLOCK)
        s_lo, s_hi = max(lo, b_lo), min(hi, b_hi)
        if s_hi <= s_lo:
            continue
        rng = np.random.default_rng([seed, 0x57, b])
        nb = b_hi - b_lo
        fp = _flow_params(rng, nb); bp = _blob_params(rng, nb); coef = _field_coeffs(rng, nb)
        cnt_t = nt_all[b_lo:b_hi]; cnt_m = nm_all[b_lo:b_hi]
        mask = _blob_masks(bp, cnt_t)
        tt, cell = np.nonzero(mask)                                          # template-major, scan order within a template
        cx = (cell % BLK_W).astype(np.int16); cy = (cell // BLK_W).astype(np.int16)
        b_tori = (-flow_direction(fp[tt], cx, cy)).astype(np.float32)        # extraction_rolled.py:125
        b_codes = np.empty((len(tt), cb.M), np.uint8)
        step = 1 << 18
        for a in range(0, len(tt), step):
            d = _descriptors(coef, tt[a:a + step], cx[a:a + step], cy[a:a + step], sigma, rng, device)
            b_codes[a:a + step] = encode(d)
        # minutiae: a random foreground cell + jitter, direction along the flow (either sense), descriptor = the field there + an own component
        mt = np.repeat(np.arange(nb), cnt_m)
        t_start = np.concatenate([[0], np.cumsum(cnt_t)])
        pick = t_start[mt] + (rng.random(len(mt)) * cnt_t[mt]).astype(np.int64)
        jx = rng.integers(-8, 8, len(mt)); jy = rng.integers(-8, 8, len(mt))
        b_mx = np.clip(24 + 16 * cx[pick].astype(np.int64) + jx, 0, IMG_W - 1).astype(np.int16)
        b_my = np.clip(24 + 16 * cy[pick].astype(np.int64) + jy, 0, IMG_H - 1).astype(np.int16)
        fx = (b_mx.astype(np.float32) - 24) / 16; fy = (b_my.astype(np.float32) - 24) / 16
        b_mo = -flow_direction(fp[mt], fx, fy) + np.float32(np.pi) * rng.integers(0, 2, len(mt)).astype(np.float32)
        b_mo = (np.mod(b_mo + np.float32(np.pi), np.float32(2 * np.pi)) - np.float32(np.pi)).astype(np.float32)
        b_des = _descriptors(coef, mt, fx, fy, MINU_SIGMA, rng, device)
        m_skip = int(cnt_m[:s_lo - b_lo].sum()); m_take = int(nm_all[s_lo:s_hi].sum())
        t_skip = int(cnt_t[:s_lo - b_lo].sum()); t_take = int(nt_all[s_lo:s_hi].sum())
        md = int(mo[s_lo - lo]); td = int(to[s_lo - lo])
        mx[md:md + m_take] = b_mx[m_skip:m_skip + m_take]; my[md:md + m_take] = b_my[m_skip:m_skip + m_take]
        mori[md:md + m_take] = b_mo[m_skip:m_skip + m_take]; des[md:md + m_take] = b_des[m_skip:m_skip + m_take]
        tx[td:td + t_take] = cx[t_skip:t_skip + t_take]; ty[td:td + t_take] = cy[t_skip:t_skip + t_take]
        tori[td:td + t_take] = b_tori[t_skip:t_skip + t_take]; codes[td:td + t_take] = b_codes[t_skip:t_skip + t_take]
    return PackedGallery(mo, mx, my, mori, des, to, tx, ty, tori, codes)


def make_structured_rolled(rng: np.random.Generator, cb: Codebook, n_minu: Optional[int] = None, n_tex: Optional[int] = None,
                           sigma: float = DUP_SIGMA[10]) -> FPTemplate:
    """One structured rolled template (small cases: tests, the CLI)."""
    if n_minu is None:
        n_minu = int(np.clip(round(rng.normal(80, 15)), 20, 200))
    if n_tex is None:
        n_tex = int(rng.integers(600, 1001))
    n_tex = min(n_tex, BLK_W * BLK_H)
    fp = _flow_params(rng, 1); bp = _blob_params(rng, 1); coef = _field_coeffs(rng, 1)
    cell = np.nonzero(_blob_masks(bp, np.array([n_tex]))[0])[0]
    cx = (cell % BLK_W).astype(np.int16); cy = (cell // BLK_W).astype(np.int16)
    zero = np.zeros(len(cell), np.int64)
    t = FPTemplate()
    codes = cb.encode_fast(_descriptors(coef, zero, cx, cy, sigma, rng)) if n_tex else np.zeros((0, cb.M), np.uint8)
    px, py = _minutiae_in_blob(rng, cell if len(cell) else np.arange(BLK_W * BLK_H), n_minu)
    fx = (px.astype(np.float32) - 24) / 16; fy = (py.astype(np.float32) - 24) / 16
    mo = -flow_direction(fp[0], fx, fy) + np.float32(np.pi) * rng.integers(0, 2, n_minu).astype(np.float32)
    mo = (np.mod(mo + np.float32(np.pi), np.float32(2 * np.pi)) - np.float32(np.pi)).astype(np.float32)
    t.minu.append(MinutiaeTemplate(px, py, mo, _descriptors(coef, np.zeros(n_minu, np.int64), fx, fy, MINU_SIGMA, rng)))
    t.tex.append(TextureTemplate(cx, cy, (-flow_direction(fp[0], cx, cy)).astype(np.float32), codes=codes))
    return t


def make_structured_latent(rng: np.random.Generator, n_minu_tpl: int = 28, n_tex_lo: int = 400, n_tex_hi: int = 1000,
                           n_minu_lo: int = 20, n_minu_hi: int = 60, sigma: float = DUP_SIGMA[10]) -> FPTemplate:
    """28 minutiae templates that are noisy views of one pool of minutiae in the foreground (directions along the flow), plus one texture template: the
    foreground's grid points in scan order, two virtual minutiae per point (ori, pi + ori: extraction_latent.py:204-205, interleaved as there), descriptors
    from two smooth fields (the two orientations see different patches)."""
    fp = _flow_params(rng, 1); bp = _blob_params(rng, 1); coef = _field_coeffs(rng, 2)
    n_tex = int(rng.integers(n_tex_lo, n_tex_hi + 1))
    n_grid = max(1, min(n_tex // 2, BLK_W * BLK_H))
    cell = np.nonzero(_blob_masks(bp, np.array([n_grid]))[0])[0]
    cx = (cell % BLK_W).astype(np.int16); cy = (cell // BLK_W).astype(np.int16)
    pool = n_minu_hi
    bx, by = _minutiae_in_blob(rng, cell, pool)
    fx = (bx.astype(np.float32) - 24) / 16; fy = (by.astype(np.float32) - 24) / 16
    bo = -flow_direction(fp[0], fx, fy) + np.float32(np.pi) * rng.integers(0, 2, pool).astype(np.float32)
    bo = (np.mod(bo + np.float32(np.pi), np.float32(2 * np.pi)) - np.float32(np.pi)).astype(np.float32)
    bd = _descriptors(coef, np.zeros(pool, np.int64), fx, fy, MINU_SIGMA, rng)
    t = FPTemplate()
    for _ in range(n_minu_tpl):
        n = int(rng.integers(n_minu_lo, n_minu_hi + 1))
        sel = np.sort(rng.permutation(pool)[:n])
        des = _unit(bd[sel] + rng.standard_normal((n, 96)) * 0.02)
        t.minu.append(MinutiaeTemplate(bx[sel].copy(), by[sel].copy(), bo[sel].copy(), des))
    go = (-flow_direction(fp[0], cx, cy)).astype(np.float32)
    tx = np.repeat(cx, 2); ty = np.repeat(cy, 2)
    to = np.stack([go, go + np.float32(np.pi)], axis=1).reshape(-1).astype(np.float32)
    which = np.tile(np.array([0, 1], np.int64), n_grid)
    td = _descriptors(coef, which, tx, ty, sigma, rng)
    t.tex.append(TextureTemplate(tx, ty, to, des=td))
    t._pool = (bx, by, bo, bd)
    return t


def make_structured_mate(rng: np.random.Generator, cb: Codebook, latent: FPTemplate, frac: float = 0.75, noise: float = 0.08,
                         n_minu: Optional[int] = None, n_tex: Optional[int] = None, sigma: float = DUP_SIGMA[10]) -> FPTemplate:
    """A structured rolled template that shares `frac` of the latent's minutiae pool (translated by pixels) and, on the cells that are a whole-block translation of
    `frac` of the latent's grid points, the latent's orientation and descriptors (first orientation of each point; codes = encode(descriptor + noise / 2))."""
    t = make_structured_rolled(rng, cb, n_minu, n_tex, sigma)
    bx, by, bo, bd = latent._pool
    m = t.minu[0]
    k = min(int(len(bx) * frac), m.n)
    sel = rng.permutation(len(bx))[:k]
    dx, dy = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
    m.x[:k] = np.clip(bx[sel] + dx, 0, IMG_W - 1); m.y[:k] = np.clip(by[sel] + dy, 0, IMG_H - 1)
    m.ori[:k] = bo[sel]
    m.des[:k] = _unit(bd[sel] + rng.standard_normal((k, 96)) * noise)
    lt = latent.tex[0]; rt = t.tex[0]
    first = np.arange(0, lt.n, 2)                                           # one orientation per grid point in a rolled print
    bdx, bdy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
    lcell = (lt.y[first].astype(np.int64) + bdy) * BLK_W + (lt.x[first].astype(np.int64) + bdx)
    inside = (lt.x[first] + bdx >= 0) & (lt.x[first] + bdx < BLK_W) & (lt.y[first] + bdy >= 0) & (lt.y[first] + bdy < BLK_H)
    rcell = rt.y.astype(np.int64) * BLK_W + rt.x.astype(np.int64)
    pos = {int(c): i for i, c in enumerate(rcell)}
    both = [(pos[int(c)], int(f)) for c, f, ok in zip(lcell, first, inside) if ok and int(c) in pos]
    rng.shuffle(both)
    both = both[:int(len(first) * frac)]
    if both:
        ri = np.array([a for a, _ in both]); li = np.array([b for _, b in both])
        rt.ori[ri] = lt.ori[li]
        rt.codes[ri] = cb.encode_fast(lt.des[li] + rng.standard_normal((len(li), 96)).astype(np.float32) * (noise * 0.5))
    return t


def make_structured_latents(seed: int, n: int, **kw) -> List[FPTemplate]:
    return [make_structured_latent(np.random.default_rng([seed, 0x58, i]), **kw) for i in range(n)]


def plant_structured_mates(seed: int, gal: PackedGallery, cb: Codebook, latents: List[FPTemplate], G: Optional[int] = None, lo: int = 0,
                           n_partial: int = 3, sigma: float = DUP_SIGMA[10]) -> Dict[int, List[Tuple[int, float]]]:
    """synth.plant_mates for the structured gallery: the same slots, structured mates of the slot's own size."""
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
            rng = np.random.default_rng([seed, 0x59, q, r])
            gal.set_template(k, make_structured_mate(rng, cb, L, frac=frac, noise=0.08, n_minu=nm, n_tex=nt, sigma=sigma))
    return planted


def dup_share(codes: np.ndarray, off: np.ndarray) -> float:
    """Share of texture points whose 16-byte code vector also occurs at ANOTHER point of the same template."""
    codes = np.ascontiguousarray(codes)
    key = codes.view([("a", "u8"), ("b", "u8")]).reshape(-1) if codes.shape[1] == 16 else None
    n_dup = 0
    for t in range(len(off) - 1):
        a, b = int(off[t]), int(off[t + 1])
        if b - a < 2:
            continue
        _, inv, cnt = np.unique(key[a:b] if key is not None else codes[a:b], axis=0, return_inverse=True, return_counts=True)
        n_dup += int((cnt[inv.reshape(-1)] > 1).sum())
    return n_dup / max(1, int(off[-1] - off[0]))
