#!/usr/bin/env python
"""Off-envelope parity sweep (not part of the test suite; tests/test_gpu_parity.py runs a 200-pair slice of it): latents and rolled prints with the shapes the
reference's reader accepts but SURVEY section 8d's synthetic envelope never produces — rolled minutiae templates of 129 .. 2000 minutiae, latent ones of 65 .. 200,
texture templates of 1001 .. 1900 rows (the scorer's clamp, matcher.cpp:544-547), pixel coordinates on both sides of 2047 (msu-latentafis_amd/host/synth.py:
make_offenvelope_set).  Every per-part score and the fused score of every pair against the oracle (tie_mode 1), bit for bit; the candidate kernel's task routing
(shape classes / any-shape fallback) is reported with it.   usage: python tools/offenv_sweep.py [seed] [n_latents] [n_rolled] [out.json]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 80
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 250
out_path = sys.argv[4] if len(sys.argv) > 4 else ""
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
t0 = time.time()
lats, rolled, mates = S.make_offenvelope_set(seed, NL, NR, cb)
m = M.Matcher(cbb)
for R in rolled: m.gallery_add_dat(T.write_rolled(R))
m.gallery_commit(0)
routing = {k: 0 for k in ("minu_tasks", "minu_tasks_small", "minu_tasks_medium", "minu_tasks_large", "minu_fallback_tasks")}
got_parts, got_scores = [], []
for q0 in range(0, NL, 16):                                                       # a few latents per call: the routing counters are per call
    r = m.search(lats[q0:q0 + 16], k=0, want_parts=True)
    got_parts.append(r["parts"]); got_scores.append(r["scores"])
    tm = m.timing()
    for k in routing: routing[k] += int(tm.get(k, 0))
parts_g = np.concatenate(got_parts); scores_g = np.concatenate(got_scores)
m.close()
t_gpu = time.time() - t0
orc = Oracle(); ocb = orc.codebook(cbb)
hr = [orc.rolled(T.write_rolled(R))[0] for R in rolled]
bad = nz = mate_pos = 0
first = []
for qi, L in enumerate(lats):
    hl, _ = orc.latent(ocb, T.write_latent(L))
    rc, sc, parts = orc.search(ocb, hl, hr, tie_mode=1, threads=orc.lib.orc_num_threads(), want_parts=True)
    g = np.concatenate([parts_g[qi], scores_g[qi][:, None]], axis=1)
    d = (g.view(np.uint32) != parts.view(np.uint32)).any(axis=1)
    bad += int(d.sum()); nz += int((parts[:, :4] > 0).sum()); mate_pos += sum(1 for gi in mates[qi] if sc[gi] > 0)
    if d.any() and len(first) < 5:
        gi = int(np.argwhere(d)[0, 0])
        first.append({"latent": qi, "rolled": gi, "latent_minutiae": [L.minu[i].n for i in (26, 2, 11)], "rolled_minutiae": rolled[gi].minu[0].n, "got": g[gi].tolist(), "want": parts[gi].tolist()})
    orc.lib.orc_latent_free(hl)
for h in hr: orc.lib.orc_rolled_free(h)
res = {"seed": seed, "latents": NL, "rolled": NR, "pairs": NL * NR, "non_zero_part_scores": nz, "mates_with_positive_score": mate_pos, "pairs_with_any_differing_bit": bad,
       "candidate_task_routing": routing, "rolled_minutiae_counts": sorted({R.minu[0].n for R in rolled}), "latent_minutiae_counts": sorted({L.minu[i].n for L in lats for i in (26, 2, 11)}),
       "latent_texture_rows": [int(min(L.tex[0].n for L in lats)), int(max(L.tex[0].n for L in lats))], "max_pixel_coordinate": int(max(max(int(R.minu[0].x.max()), int(R.minu[0].y.max())) for R in rolled)),
       "seconds": {"generate_and_gpu": round(t_gpu, 1), "total": round(time.time() - t0, 1)}, "first_mismatches": first}
print(json.dumps(res))
if out_path:
    with open(out_path, "w") as f: json.dump(res, f, indent=1)
sys.exit(1 if bad else 0)
