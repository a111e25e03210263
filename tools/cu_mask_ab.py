"""Same library, same workload, ONE process: contexts whose stream was created with different CU masks (AFIS_CU_MASK is read at afis_create), interleaved steps.
python tools/cu_mask_ab.py G Q name=hexwords ...   (name 'none' = no mask)"""
import sys, importlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
G, Q = int(sys.argv[1]), int(sys.argv[2]); specs = sys.argv[3:]
lats = S.make_latents(1, Q); gal = S.make_packed_gallery(1, G, cb); S.plant_mates(1, gal, cb, lats)
ms = []
for sp in specs:
    name, _, m = sp.partition("=")
    if m: os.environ["AFIS_CU_MASK"] = m
    else: os.environ.pop("AFIS_CU_MASK", None)
    mt = M.Matcher(cbb); mt.gallery_add_packed(gal); mt.gallery_commit(0)
    ms.append((name, mt, mt.upload_queries(lats)))
best = {}
for rep in range(4):
    for name, mt, qh in ms:
        r = mt.search_resident(qh); tm = mt.timing()
        if rep: best[name] = tm if name not in best else {k: min(best[name][k], v) if k.endswith("_ms") else v for k, v in tm.items()}
for name, b in best.items():
    print(name, {k: round(v, 2) for k, v in b.items() if k in ("adc_bound_ms", "adc_refine_ms", "tex_tail_ms", "cands_ms", "minu_graph_ms", "total_ms")}, flush=True)
