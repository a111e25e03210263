#!/usr/bin/env python
"""How far can a score move when the sums the reference leaves to Eigen are taken in another order?

The reference's only third-party arithmetic on this path is Eigen's: the descriptor product of S1 (matching/matcher.cpp:443), the row / column
sums of S2 (:455-470) and the mat-vec + sum of the S8 power iteration (:1279-1289, :1401-1411).  Eigen is neither in the reference tree nor in this
image, so the oracle (and the HIP path) fix ONE order, and the claim "inside SURVEY section 8d's tolerance" needs a number.  This tool produces it on the
CPU: the oracle evaluates every pair in its canonical order (sum_order 0) and in each of the orders an Eigen build could take (oracle/afis_oracle.cpp,
"accumulation orders": 1 = unfused k-ascending GEMM — the reference's own build flags; 2 = that + 4-lane vectorised reductions; 3 = 4 lanes everywhere;
4 = 8 lanes + FMA; 5 = pairwise), in both tie modes, and counts: positive pairs with a differing bit, pairs beyond 1e-3 * max(1, |s|), changes of the
top-24 over strictly positive scores, rank-1 changes.

usage: python tools/order_sweep.py [--workload headline|structured] [--seed 7] [--queries 32] [--gallery 3000] [--out profiles/r06_order_sweep.json]
"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth")
SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")

ORDERS = {1: "S1 unfused k-ascending (GEBP of the reference's own flags: -O3, SSE2, no FMA); other sums sequential",
          2: "S1 unfused k-ascending; row sums, mat-vec rows and sum(c) as 4 strided partial sums (Eigen's SSE redux / row-major gemv); column sums sequential",
          3: "every sum incl. S1 as 4 strided partial sums, unfused", 4: "every sum as 8 strided partial sums with FMA (an AVX2 + FMA build)", 5: "every sum pairwise, unfused"}


def make_set(workload, seed, Q, G, cb, dup=10):
    if workload == "structured":
        sg = SS.DUP_SIGMA[dup]
        lats = SS.make_structured_latents(seed, Q, sigma=sg)
        import scipy.cluster.vq as vq
        def enc(d):
            out = np.empty((len(d), 16), np.uint8)
            for m in range(16): out[:, m] = vq.vq(d[:, 6 * m:6 * m + 6].astype(np.float64), cb.words[m].astype(np.float64))[0]
            return out
        gal = SS.make_packed_gallery_structured(seed, G, cb, sigma=sg, encode=enc)
        planted = SS.plant_structured_mates(seed, gal, cb, lats, G=G, sigma=sg)
    else:
        lats = S.make_latents(seed, Q); gal = S.make_packed_gallery(seed, G, cb); planted = S.plant_mates(seed, gal, cb, lats, G=G)
    return lats, gal, planted


def compare(s0, p0, s1, p1, acc, k=24):
    G = len(s0)
    pos = (s0 > 0) | (s1 > 0)
    bit = s0.view(np.uint32) != s1.view(np.uint32)
    tol = 1e-3 * np.maximum(1.0, np.abs(s0))
    far = np.abs(s0 - s1) > tol
    acc["pairs"] += G; acc["positive_pairs"] += int(pos.sum()); acc["positive_pairs_any_bit"] += int((pos & bit).sum()); acc["pairs_beyond_1e-3"] += int(far.sum())
    acc["max_abs_diff"] = max(acc["max_abs_diff"], float(np.abs(s0 - s1).max()))
    rel = np.abs(s0 - s1) / np.maximum(1.0, np.abs(s0))
    acc["max_rel_diff_within_tolerance"] = max(acc["max_rel_diff_within_tolerance"], float(rel[~far].max()) if (~far).any() else 0.0)
    for c, name in enumerate(("minutiae_26", "minutiae_2", "minutiae_11", "texture")):
        a, b = p0[:, c], p1[:, c]
        acc["part_" + name + "_any_bit"] += int((a.view(np.uint32) != b.view(np.uint32)).sum())
        acc["part_" + name + "_beyond_1e-3"] += int((np.abs(a - b) > 1e-3 * np.maximum(1.0, np.abs(a))).sum())
    r0 = np.lexsort((np.arange(G), -s0))[:k]; r1 = np.lexsort((np.arange(G), -s1))[:k]
    n_pos = int(min((s0[r0] > 0).sum(), (s1[r1] > 0).sum()))
    # identical top-24 sets and order over strictly positive, NON-TIED scores: positions whose score equals a neighbour's in either list are skipped
    def untied(s, r): v = s[r]; t = np.zeros(len(r), bool); t[1:] |= v[1:] == v[:-1]; t[:-1] |= v[:-1] == v[1:]; return ~t
    ok = untied(s0, r0)[:n_pos] & untied(s1, r1)[:n_pos]
    acc["top24_changes_over_positive_untied"] += int(not np.array_equal(r0[:n_pos][ok], r1[:n_pos][ok]))
    acc["top24_set_changes"] += int(set(r0[:n_pos].tolist()) != set(r1[:n_pos].tolist()))
    acc["rank1_changes"] += int(r0[0] != r1[0])
    return far


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="headline", choices=["headline", "structured"]); ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--queries", type=int, default=32); ap.add_argument("--gallery", type=int, default=3000); ap.add_argument("--dup", type=int, default=10)
    ap.add_argument("--orders", default="1,2,3,4,5"); ap.add_argument("--out", default="")
    a = ap.parse_args()
    cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
    lats, gal, planted = make_set(a.workload, a.seed, a.queries, a.gallery, cb, a.dup)
    orc = Oracle(); ocb = orc.codebook(cbb); nt = orc.lib.orc_num_threads()
    hr = [orc.rolled(T.write_rolled(gal.template(g)))[0] for g in range(gal.G)]
    orders = [int(x) for x in a.orders.split(",")]
    keys = ["pairs", "positive_pairs", "positive_pairs_any_bit", "pairs_beyond_1e-3", "top24_changes_over_positive_untied", "top24_set_changes", "rank1_changes"] + \
           ["part_" + n + s for n in ("minutiae_26", "minutiae_2", "minutiae_11", "texture") for s in ("_any_bit", "_beyond_1e-3")]
    res = {(tie, so): dict({k_: 0 for k_ in keys}, max_abs_diff=0.0, max_rel_diff_within_tolerance=0.0, examples=[]) for tie in (0, 1) for so in orders}
    t0 = time.time()
    for qi, L in enumerate(lats):
        hl, _ = orc.latent(ocb, T.write_latent(L))
        for tie in (0, 1):
            _, s0, p0 = orc.search(ocb, hl, hr, tie_mode=tie, threads=nt, want_parts=True)
            for so in orders:
                _, s1, p1 = orc.search(ocb, hl, hr, tie_mode=tie | (so << 4), threads=nt, want_parts=True)
                acc = res[(tie, so)]
                far = compare(s0, p0, s1, p1, acc)
                for g in np.argwhere(far).ravel()[:2]:
                    if len(acc["examples"]) < 6:
                        acc["examples"].append({"query": qi, "gallery": int(g), "planted": bool(int(g) in [x for x, _ in planted[qi]]), "canonical": [float(v) for v in p0[g]], "this_order": [float(v) for v in p1[g]]})
        orc.lib.orc_latent_free(hl)
        print(f"latent {qi + 1}/{len(lats)}  {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    out = {"what": __doc__.split("\n\n")[0].strip(), "workload": a.workload + (f" (dup {a.dup})" if a.workload == "structured" else ""), "seed": a.seed, "queries": a.queries, "gallery": a.gallery,
           "pairs_per_order_and_tie_mode": a.queries * a.gallery, "threads": nt, "oracle_seconds": round(time.time() - t0, 1), "orders": {str(k_): v for k_, v in ORDERS.items() if k_ in orders}, "results": []}
    for (tie, so), acc in sorted(res.items()):
        acc = dict(acc)
        acc["share_of_pairs_within_1e-3"] = round(1.0 - acc["pairs_beyond_1e-3"] / max(1, acc["pairs"]), 6)
        acc["share_of_positive_pairs_within_1e-3"] = round(1.0 - acc["pairs_beyond_1e-3"] / max(1, acc["positive_pairs"]), 6)
        out["results"].append(dict(tie_mode=tie, sum_order=so, **acc))
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
