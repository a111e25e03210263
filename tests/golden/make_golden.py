#!/usr/bin/env python
"""Generates tests/golden/golden_pairs.npz — small seeded (latent, rolled) .dat inputs and the per-pair scores the ORACLE
(oracle/afis_oracle.cpp) gives for them, in both tie modes, plus hashes of the S4 look-up tables and S5/S6 row maxima.

What these vectors pin: the oracle against itself over time (regressions) and the HIP path against the oracle on fixed inputs.
They are NOT outputs of the reference: matching/matcher.cpp is unbuildable in this image (Eigen / Boost absent).
The only stage checked against reference code is S4 / the template data model, through oracle/_ref (tests/test_oracle.py).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402

T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")


def build_inputs(cb):
    rng = np.random.default_rng(20240928)
    lats = [S.make_latent(rng, n_tex_lo=210, n_tex_hi=250, n_minu_lo=8, n_minu_hi=16) for _ in range(2)]
    gal = []
    for L in lats:
        gal.append(S.make_mate(rng, cb, L, frac=0.8, n_minu=30, n_tex=300))
        gal.append(S.make_mate(rng, cb, L, frac=0.5, n_minu=34, n_tex=280))
        gal.append(S.make_mate(rng, cb, L, frac=0.3, n_minu=26, n_tex=320))
    for _ in range(6):
        gal.append(S.make_rolled(rng, cb, n_minu=int(rng.integers(20, 40)), n_tex=int(rng.integers(250, 330))))
    return lats, gal


def main():
    with open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb") as f:
        cbb = f.read()
    cb = T.Codebook.from_bytes(cbb)
    lats, gal = build_inputs(cb)
    orc = Oracle(); ocb = orc.codebook(cbb)
    out = {}
    lat_dat = [T.write_latent(L) for L in lats]; rol_dat = [T.write_rolled(R) for R in gal]
    for i, b in enumerate(lat_dat): out[f"latent_{i}"] = np.frombuffer(b, np.uint8)
    for j, b in enumerate(rol_dat): out[f"rolled_{j}"] = np.frombuffer(b, np.uint8)
    hl = [orc.latent(ocb, b)[0] for b in lat_dat]; hr = [orc.rolled(b)[0] for b in rol_dat]
    parts = np.zeros((2, len(hl), len(hr), 5), np.float32)       # [tie_mode][latent][rolled][s0,s1,s2,tex,final]
    for tm in (0, 1):
        for i, h in enumerate(hl):
            rc, sc, p = orc.search(ocb, h, hr, tie_mode=tm, want_parts=True)
            assert rc == 0
            parts[tm, i] = p
    out["parts"] = parts
    lut_sha = []
    for i, h in enumerate(hl):
        lut_sha.append(hashlib.sha256(orc.lut(h, 0).tobytes()).hexdigest())
    out["lut_sha256"] = np.array(lut_sha)
    rm_sha = []
    for i, h in enumerate(hl):
        for j, r in enumerate(hr):
            v, a = orc.texture_rowmax(ocb, h, r)
            rm_sha.append(hashlib.sha256(v.tobytes() + a.astype(np.int32).tobytes()).hexdigest())
    out["rowmax_sha256"] = np.array(rm_sha)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_pairs.npz"), **out)
    print("wrote golden_pairs.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith(("latent_", "rolled_"))})
    print("final scores (tie_mode 1):\n", np.round(parts[1, :, :, 4], 3))


if __name__ == "__main__":
    main()
