"""A/B of option settings of ONE libafis_hip.so inside one process on one box: python tools/opt_ab.py G Q name=v[,name=v] name=v ...
Every setting scores the same workload, interleaved, 4 rounds (first = warm-up); prints the minimum stage times per setting and whether the
scores equal the first setting's bit for bit."""
import sys, importlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
G, Q = int(sys.argv[1]), int(sys.argv[2]); settings = sys.argv[3:]
lats = S.make_latents(1, Q); gal = S.make_packed_gallery(1, G, cb); S.plant_mates(1, gal, cb, lats)
m = M.Matcher(cbb); m.gallery_add_packed(gal); m.gallery_commit(0)
qh = m.upload_queries(lats)
ref = None; best = [None] * len(settings)
for rep in range(4):
    for i, st in enumerate(settings):
        for kv in st.split(","):
            m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        r = m.search_resident(qh, want_scores=True); tm = m.timing()
        if ref is None: ref = r["scores"]
        same = bool((r["scores"] == ref).all())
        if rep:
            best[i] = tm if best[i] is None else {k: min(best[i][k], v) if k.endswith("_ms") else v for k, v in tm.items()}
            best[i]["identical"] = same
for st, b in zip(settings, best):
    print(st, {k: round(v, 2) for k, v in b.items() if k.endswith("_ms")}, "identical", b["identical"])
