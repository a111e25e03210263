#!/usr/bin/env python
"""How far is "ties by ascending index" (the HIP path, oracle tie_mode=1) from the order the reference binary executes
(libstdc++ std::sort on a non-strict key, oracle tie_mode=0; matching/matcher.cpp:476, :741, :1301, :1423, :1590, :306)?

CPU only: both orders are modes of the oracle.  Q latents x G synthetic gallery templates at bench shapes (about 40 x 80
minutiae, 670 x 800 texture points) with planted mates; every per-part and fused score of every pair in both modes.
usage: python tools/tie_sweep.py [seed] [Q] [G] [n_partial] [out.json]
"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth")


def sweep(seed, Q, G, n_partial, k=24):
    cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
    lats = S.make_latents(seed, Q); gal = S.make_packed_gallery(seed, G, cb); planted = S.plant_mates(seed, gal, cb, lats, G=G, n_partial=n_partial)
    orc = Oracle(); ocb = orc.codebook(cbb); nt = orc.lib.orc_num_threads()
    hr = [orc.rolled(T.write_rolled(gal.template(g)))[0] for g in range(G)]
    out = {"seed": seed, "queries": Q, "gallery": G, "planted_per_query": 1 + n_partial, "pairs": Q * G,
           "positive_pairs": 0, "positive_pairs_any_bit": 0, "positive_pairs_beyond_1e-3": 0, "pairs_any_bit": 0,
           "nonzero_part_scores": 0, "part_scores_any_bit": 0, "part_scores_beyond_1e-3": 0,
           "planted_pairs": 0, "planted_any_bit": 0, "planted_beyond_1e-3": 0,
           "top24_set_changes": 0, "top24_order_changes_over_positive": 0, "rank1_changes": 0, "worst": []}
    t0 = time.time()
    for qi, L in enumerate(lats):
        hl, _ = orc.latent(ocb, T.write_latent(L))
        _, s0, p0 = orc.search(ocb, hl, hr, tie_mode=0, threads=nt, want_parts=True)
        _, s1, p1 = orc.search(ocb, hl, hr, tie_mode=1, threads=nt, want_parts=True)
        pos = (s0 > 0) | (s1 > 0)
        bit = s0.view(np.uint32) != s1.view(np.uint32)
        far = np.abs(s0 - s1) > 1e-3 * np.maximum(1.0, np.abs(s0))
        out["positive_pairs"] += int(pos.sum()); out["positive_pairs_any_bit"] += int((pos & bit).sum())
        out["positive_pairs_beyond_1e-3"] += int((pos & far).sum()); out["pairs_any_bit"] += int(bit.sum())
        a, b = p0[:, :4], p1[:, :4]
        nzp = (a > 0) | (b > 0)
        out["nonzero_part_scores"] += int(nzp.sum()); out["part_scores_any_bit"] += int((a.view(np.uint32) != b.view(np.uint32)).sum())
        out["part_scores_beyond_1e-3"] += int((np.abs(a - b) > 1e-3 * np.maximum(1.0, np.abs(a))).sum())
        pl = np.array([g for g, _ in planted[qi]])
        out["planted_pairs"] += len(pl); out["planted_any_bit"] += int(bit[pl].sum()); out["planted_beyond_1e-3"] += int(far[pl].sum())
        # rank lists: score descending, index ascending (the documented rule) in both modes
        r0 = np.lexsort((np.arange(G), -s0))[:k]; r1 = np.lexsort((np.arange(G), -s1))[:k]
        out["top24_set_changes"] += int(set(r0.tolist()) != set(r1.tolist()))
        n_pos = int(min((s0[r0] > 0).sum(), (s1[r1] > 0).sum()))
        out["top24_order_changes_over_positive"] += int(not np.array_equal(r0[:n_pos], r1[:n_pos]))
        out["rank1_changes"] += int(r0[0] != r1[0])
        for g in np.argwhere(far).ravel()[:4]:
            out["worst"].append({"query": qi, "gallery": int(g), "planted": bool(g in pl), "tie0": [float(v) for v in p0[g]], "tie1": [float(v) for v in p1[g]]})
        orc.lib.orc_latent_free(hl)
    out["worst"] = out["worst"][:12]
    out["frac_positive_any_bit"] = out["positive_pairs_any_bit"] / max(1, out["positive_pairs"])
    out["frac_positive_beyond_1e-3"] = out["positive_pairs_beyond_1e-3"] / max(1, out["positive_pairs"])
    out["frac_all_pairs_beyond_1e-3"] = out["positive_pairs_beyond_1e-3"] / max(1, out["pairs"])
    out["oracle_seconds"] = round(time.time() - t0, 1)
    return out


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    n_partial = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    r = sweep(seed, Q, G, n_partial)
    print(json.dumps(r, indent=1))
    if len(sys.argv) > 5:
        json.dump(r, open(sys.argv[5], "w"), indent=1)
