#!/usr/bin/env python
"""Wide parity sweep (not part of the test suite): Q latents x G synthetic gallery templates, every per-part score and the fused
score of every pair against the oracle (all host threads), bit for bit in tie_mode=1 (equal keys by ascending index, what the HIP
path implements) and, with a 4th argument, against tie_mode=0 (libstdc++ std::sort order, what the reference binary executes):
positive pairs with any differing bit / beyond 1e-3, top-24 changes.  usage: python tools/parity_sweep.py [seed] [Q] [G] [tie0]
(tools/tie_sweep.py does the tie_mode 0-vs-1 comparison on the CPU alone.)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
wl = S.WORKLOADS[os.environ.get("AFIS_SWEEP_WORKLOAD", "headline")]                 # AFIS_SWEEP_WORKLOAD=wide: the off-envelope shapes of bench.py --workload wide
m = M.Matcher(cbb)
TIE = 1
if os.environ.get("AFIS_SWEEP_S3_ORDER") == "1":                                     # option s3_tie_order 1 against the oracle's tie mode 4 (std::sort at S3, the stable order elsewhere)
    m.set_option("s3_tie_order", 1); TIE = 4
if os.environ.get("AFIS_SWEEP_S3_ORDER") == "2":                                     # option ref_tie_order 2 against tie mode 9 (std::sort at S3, S8 and S9)
    m.set_option("ref_tie_order", 2); TIE = 9
if os.environ.get("AFIS_SWEEP_WORKLOAD") == "structured":                            # AFIS_SWEEP_WORKLOAD=structured [AFIS_SWEEP_DUP=0|10|30]: templates with the structure of extracted prints (host/synth_structured.py)
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured"); sg = SS.DUP_SIGMA[int(os.environ.get("AFIS_SWEEP_DUP", "10"))]
    if os.environ.get("AFIS_SWEEP_IDENTITY"): SS.IDENTITY_WEIGHT = float(os.environ["AFIS_SWEEP_IDENTITY"])       # 1.0: a twelfth of the minutiae lists is short of 120 positive similarities
    lats = SS.make_structured_latents(seed, Q, sigma=sg); gal = SS.make_packed_gallery_structured(seed, G, cb, sigma=sg, encode=m.pq_encode); SS.plant_structured_mates(seed, gal, cb, lats, G=G, sigma=sg)
    print("structured: share of texture points whose code vector occurs twice in their template:", round(SS.dup_share(gal.tex_codes, gal.tex_off), 4))
else:
    lats = S.make_latents(seed, Q, **wl["latent"]); gal = S.make_packed_gallery(seed, G, cb, **wl["gallery"]); S.plant_mates(seed, gal, cb, lats, G=G)
m.gallery_add_packed(gal); m.gallery_commit(0)
if os.environ.get("AFIS_ADC_VARIANT"): m.set_option("adc_variant", int(os.environ["AFIS_ADC_VARIANT"]))
r = m.search(lats, k=0, want_parts=True)
orc = Oracle(); ocb = orc.codebook(cbb)
hr = [orc.rolled(T.write_rolled(gal.template(g)))[0] for g in range(G)]
TIE0 = len(sys.argv) > 4
t0 = time.time(); bad = 0; nz = 0; pos0 = bit0 = far0 = top0 = 0
for qi, L in enumerate(lats):
    hl, _ = orc.latent(ocb, T.write_latent(L))
    rc, sc, parts = orc.search(ocb, hl, hr, tie_mode=TIE, threads=orc.lib.orc_num_threads(), want_parts=True)
    got = np.concatenate([r["parts"][qi], r["scores"][qi][:, None]], axis=1)
    diff = got.view(np.uint32) != parts.view(np.uint32)
    bad += int(diff.any(axis=1).sum()); nz += int((parts[:, :4] > 0).sum())
    if diff.any():
        g = int(np.argwhere(diff.any(axis=1))[0, 0]); print("first mismatch: query", qi, "gallery", g, "got", got[g], "want", parts[g])
    if TIE0:
        rc, s0, p0 = orc.search(ocb, hl, hr, tie_mode=0, threads=orc.lib.orc_num_threads(), want_parts=True)
        gs = r["scores"][qi]
        pos = (s0 > 0) | (gs > 0)
        pos0 += int(pos.sum()); bit0 += int((pos & (s0.view(np.uint32) != gs.view(np.uint32))).sum())
        far0 += int((np.abs(s0 - gs) > 1e-3 * np.maximum(1.0, np.abs(s0))).sum())
        a = np.lexsort((np.arange(G), -s0.astype(np.float64)))[:24]; b = np.lexsort((np.arange(G), -gs.astype(np.float64)))[:24]
        npos = int(min((s0[a] > 0).sum(), (gs[b] > 0).sum()))
        top0 += int(not np.array_equal(a[:npos], b[:npos]))
tmr = m.timing()
npos_pairs = int((r["scores"] > 0).sum())
print(f"seed {seed}: {Q} x {G} pairs, {nz} non-zero part scores, {npos_pairs} pairs with a positive score, pairs with any differing bit: {bad}  (oracle {time.time() - t0:.1f} s)")
print("SWEEP_JSON " + __import__("json").dumps({"workload": os.environ.get("AFIS_SWEEP_WORKLOAD", "headline"), "oracle_tie_mode": TIE, "ref_tie_order": {1: 0, 4: 1, 9: 2}[TIE], "identity_weight": os.environ.get("AFIS_SWEEP_IDENTITY"), "seed": seed, "Q": Q, "G": G, "pairs": Q * G, "non_zero_part_scores": nz, "pairs_with_a_positive_score": npos_pairs, "dup": os.environ.get("AFIS_SWEEP_DUP"), "pairs_with_any_differing_bit": bad,
      "candidate_task_routing": {k: int(v) for k, v in tmr.items() if k.startswith("minu_") and k.endswith("tasks")}}))
if TIE0:
    print(f"  vs tie_mode=0 (reference sort order): {pos0} positive pairs, {bit0} with a differing bit, {far0} beyond 1e-3, queries whose positive top-24 order changes: {top0}")
    print("TIE0_JSON " + __import__("json").dumps({"seed": seed, "pairs": Q * G, "positive_pairs": pos0, "positive_pairs_with_a_differing_bit": bit0, "pairs_beyond_1e-3": far0, "queries_whose_positive_top24_changes": top0}))
