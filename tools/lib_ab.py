import sys, importlib, os, glob
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"),"rb").read(); cb = T.Codebook.from_bytes(cbb)
G, Q = 10000, 4
lats = S.make_latents(1, Q); gal = S.make_packed_gallery(1, G, cb); S.plant_mates(1, gal, cb, lats)
ref = None
for path in [M.LIB_PATH] + sorted(sys.argv[1:]):
    m = M.Matcher(cbb, lib_path=path); m.gallery_add_packed(gal); m.gallery_commit(0); qh = m.upload_queries(lats)
    m.search_resident(qh); r = m.search_resident(qh, want_scores=True); tm = m.timing()
    if ref is None: ref = r["scores"]
    print(os.path.basename(path), {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}, "identical", bool((r["scores"] == ref).all()))
    m.close()
