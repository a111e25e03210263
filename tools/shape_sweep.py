#!/usr/bin/env python
"""Parity sweep over template SHAPES (not part of the test suite): latents and rolled prints with few or many minutiae (2 .. 120) and
texture points (20 .. 1000), so that the correspondence lists take every length — short lists, lists whose last block of 64 rows is
nearly empty (the graph kernels' grouped lanes), lists below and above the top-120 / top-200 cuts — every per-part score of every pair
against the oracle (tie_mode 1), bit for bit.   usage: python tools/shape_sweep.py [seed] [n_latents] [n_rolled]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 24
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
rng = np.random.default_rng([seed, 99])
orc = Oracle(); ocb = orc.codebook(cbb)
bad = pairs = nz = 0
shapes = []
for li in range(NL):
    lo = int(rng.integers(2, 50)); hi = lo + int(rng.integers(0, 30))
    tl = int(rng.choice([40, 120, 199, 200, 201, 260, 330, 520, 1000])); th = tl + int(rng.integers(0, 40))
    L = S.make_latent(rng, n_tex_lo=tl, n_tex_hi=th, n_minu_lo=lo, n_minu_hi=hi)
    m = M.Matcher(cbb)
    rolled = []
    for ri in range(NR):
        nm = int(rng.choice([2, 3, 5, 9, 17, 33, 47, 64, 65, 80, 100, 120])) + int(rng.integers(0, 3))
        nt = int(rng.choice([20, 63, 64, 65, 150, 400, 800, 1000]))
        R = S.make_mate(rng, cb, L, frac=float(rng.uniform(0.3, 0.9)), n_tex=nt) if ri % 3 == 0 else S.make_rolled(rng, cb, n_minu=nm, n_tex=nt)
        rolled.append(R); m.gallery_add_dat(T.write_rolled(R))
    m.gallery_commit(0)
    got = m.search([L], k=0, want_parts=True)
    m.close()
    hl, _ = orc.latent(ocb, T.write_latent(L))
    hr = [orc.rolled(T.write_rolled(R))[0] for R in rolled]
    rc, sc, parts = orc.search(ocb, hl, hr, tie_mode=1, threads=orc.lib.orc_num_threads(), want_parts=True)
    g = np.concatenate([got["parts"][0], got["scores"][0][:, None]], axis=1)
    d = (g.view(np.uint32) != parts.view(np.uint32)).any(axis=1)
    bad += int(d.sum()); pairs += NR; nz += int((parts[:, :4] > 0).sum())
    if d.any(): print("latent", li, "minutiae", lo, hi, "texture", len(L.tex[0].x), "differs at rolled", np.argwhere(d).ravel()[:8], g[d][:2], parts[d][:2])
    for h in hr: orc.lib.orc_rolled_free(h)
    orc.lib.orc_latent_free(hl)
print(f"seed {seed}: {pairs} pairs over {NL} latent shapes, {nz} non-zero part scores, pairs with any differing bit: {bad}")
sys.exit(1 if bad else 0)
