"""A/B of several builds of libafis_hip.so inside ONE process on ONE box (box-to-box clock differences are several per cent, larger than
most kernel changes): python tools/lib_ab.py [G] [Q] <other .so> ...   — every library scores the same workload `reps` times,
interleaved; prints the minimum stage times and whether the scores are bit-identical to the first library's."""
import sys, importlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
args = sys.argv[1:]
G = int(args.pop(0)) if args and args[0].isdigit() else 20000
Q = int(args.pop(0)) if args and args[0].isdigit() else 8
wl = S.WORKLOADS[os.environ.get("AFIS_AB_WORKLOAD", "headline")]                 # AFIS_AB_WORKLOAD=wide: the off-envelope shapes (synth.py)
gkw = dict(wl["gallery"]); gkw.setdefault("n_minu_mean", float(os.environ.get("AFIS_AB_MINU_MEAN", "80")))
lats = S.make_latents(1, Q, **wl["latent"]); gal = S.make_packed_gallery(1, G, cb, **gkw); S.plant_mates(1, gal, cb, lats)
paths = [M.LIB_PATH] + sorted(args)
ms = []
for path in paths:
    m = M.Matcher(cbb, lib_path=path)
    for kv in filter(None, os.environ.get("AFIS_AB_OPTS", "").split(",")):       # e.g. AFIS_AB_OPTS=adc_variant=9,mf_kernel=12
        m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    m.gallery_add_packed(gal); m.gallery_commit(0)
    ms.append((m, m.upload_queries(lats)))
ref = None; best = [None] * len(paths)
for rep in range(4):
    for i, (m, qh) in enumerate(ms):
        r = m.search_resident(qh, want_scores=True); tm = m.timing()
        if ref is None: ref = r["scores"]
        same = bool((r["scores"] == ref).all())
        if rep:                                                             # first round = warm-up
            best[i] = tm if best[i] is None else {k: min(best[i][k], v) if k.endswith("_ms") else v for k, v in tm.items()}
            best[i]["identical"] = same
for p, b in zip(paths, best):
    print(os.path.basename(p), {k: round(v, 2) for k, v in b.items() if k.endswith("_ms")}, {k: v for k, v in b.items() if k.startswith("minu_") and k.endswith("tasks") or k.endswith("_ghz")}, "identical", b["identical"])
