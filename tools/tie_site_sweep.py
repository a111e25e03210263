#!/usr/bin/env python
"""WHICH of the reference's sorts is it whose order of equal keys can move a score?

The reference sorts with std::sort on a non-strict key at S3 (candidate norms, matching/matcher.cpp:476), S7 (row maxima, :741), in the greedy selections of S8 (:1301, :1423)
and S9 (:1590), and for the rank list (:306); the HIP path orders equal keys by ascending index (oracle tie_mode 1), the reference binary by whatever libstdc++'s introsort
yields (oracle tie_mode 0: the oracle calls std::sort itself).  On i.i.d. random templates the two differ on 46 of 79 979 positive pairs (tools/tie_sweep.py).  On STRUCTURED
templates (host/synth_structured.py) far more pairs tie — this tool says where: the oracle's tie modes 2..5 take std::sort at ONE site and the stable order elsewhere.

Result (profiles/r06_tie_site_sweep.json): the only site that matters is S3, and only for (latent, rolled) minutiae pairs that keep FEWER THAN 120 POSITIVE similarities after the
clamp of matcher.cpp:447-451 — their list of 120 is filled up with zero-norm entries, all tied, and which of them std::sort puts first is an accident of introsort.  S9's
many exact ties (uniform start vector, boolean H) do not matter: lists of <= 16 entries are insertion-sorted by libstdc++ (stable), longer ones tie among entries that end the same way.
usage: python tools/tie_site_sweep.py [--queries 4] [--gallery 1500] [--out profiles/r06_tie_site_sweep.json]
"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
SITES = {1: "ascending index at every site (what the HIP path implements)", 2: "std::sort at S9 only", 3: "std::sort at S8 and S9", 4: "std::sort at S3 only", 5: "std::sort at S7 only"}


def run(idw, dup, Q, G, seed=77):
    cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
    SS.IDENTITY_WEIGHT = idw
    sg = SS.DUP_SIGMA[dup]
    lats = SS.make_structured_latents(seed, Q, sigma=sg); gal = SS.make_packed_gallery_structured(seed, G, cb, sigma=sg); SS.plant_structured_mates(seed, gal, cb, lats, G=G, sigma=sg)
    orc = Oracle(); ocb = orc.codebook(cbb); nt = orc.lib.orc_num_threads()
    hr = [orc.rolled(T.write_rolled(gal.template(g)))[0] for g in range(G)]
    res = {t: {"pairs_beyond_1e-3": 0, "pairs_with_a_differing_bit": 0, "minutiae_parts_beyond_1e-3": 0, "texture_part_beyond_1e-3": 0, "top24_changes_over_positive": 0,
               "largest_score_among_the_pairs_beyond_1e-3": 0.0, "largest_absolute_difference": 0.0, "planted_mates_beyond_1e-3": 0} for t in SITES}
    planted = SS.plant_structured_mates(seed, gal, cb, lats, G=G, sigma=sg)          # (idempotent: the same mates again; returns their slots)
    short = tasks = pos = 0
    for qi, L in enumerate(lats):
        hl, _ = orc.latent(ocb, T.write_latent(L))
        sc = {t: orc.search(ocb, hl, hr, tie_mode=t, threads=nt, want_parts=True)[1:] for t in (0,) + tuple(SITES)}
        s0, p0 = sc[0]
        pos += int((s0 > 0).sum())
        for t in SITES:
            s1, p1 = sc[t]; r = res[t]
            far_ = np.abs(s0 - s1) > 1e-3 * np.maximum(1, np.abs(s0))
            r["pairs_beyond_1e-3"] += int(far_.sum())
            if far_.any():
                r["largest_score_among_the_pairs_beyond_1e-3"] = max(r["largest_score_among_the_pairs_beyond_1e-3"], float(np.maximum(s0, s1)[far_].max()))
                r["largest_absolute_difference"] = max(r["largest_absolute_difference"], float(np.abs(s0 - s1)[far_].max()))
                r["planted_mates_beyond_1e-3"] += int(sum(bool(far_[g_]) for g_, _f in planted[qi]))
            r["pairs_with_a_differing_bit"] += int((s0.view(np.uint32) != s1.view(np.uint32)).sum())
            r["minutiae_parts_beyond_1e-3"] += int((np.abs(p0[:, :3] - p1[:, :3]) > 1e-3 * np.maximum(1, np.abs(p0[:, :3]))).sum())
            r["texture_part_beyond_1e-3"] += int((np.abs(p0[:, 3] - p1[:, 3]) > 1e-3 * np.maximum(1, np.abs(p0[:, 3]))).sum())
            a = np.lexsort((np.arange(G), -s0))[:24]; b = np.lexsort((np.arange(G), -s1))[:24]
            n_pos = int(min((s0[a] > 0).sum(), (s1[b] > 0).sum()))
            r["top24_changes_over_positive"] += int(not np.array_equal(a[:n_pos], b[:n_pos]))
        for s_ in (26, 2, 11):
            for g in range(G):
                a_, b_ = int(gal.minu_off[g]), int(gal.minu_off[g + 1])
                short += int(((L.minu[s_].des @ gal.minu_des[a_:b_].T) > 0).sum() < 120); tasks += 1
        orc.lib.orc_latent_free(hl)
    return {"identity_weight": idw, "dup": dup, "queries": Q, "gallery": G, "pairs": Q * G, "positive_pairs": pos, "minutiae_lists": tasks, "minutiae_lists_with_fewer_than_120_positive_similarities": short,
            "against_std_sort_at_every_site": {SITES[t]: v for t, v in res.items()}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--queries", type=int, default=4); ap.add_argument("--gallery", type=int, default=1500); ap.add_argument("--out", default="")
    a = ap.parse_args()
    t0 = time.time()
    out = {"what": __doc__.split("\n\n")[0].strip(), "runs": [run(0.3, 10, a.queries, a.gallery), run(1.0, 10, a.queries, a.gallery)], "seconds": 0}
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out, indent=1))
    if a.out: json.dump(out, open(a.out, "w"), indent=1)
