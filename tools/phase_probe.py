import sys, importlib
sys.path.insert(0, "/root/repo")
import numpy as np
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open("/root/repo/tests/golden/codebook_EmbeddingSize_96_stride_16_subdim_6.dat","rb").read(); cb = T.Codebook.from_bytes(cbb)
G, Q = 10000, 4
lats = S.make_latents(1, Q); gal = S.make_packed_gallery(1, G, cb); S.plant_mates(1, gal, cb, lats)
m = M.Matcher(cbb); m.gallery_add_packed(gal); m.gallery_commit(0); qh = m.upload_queries(lats)
m.search_resident(qh); m.phase_cycles(True)
m.search_resident(qh); ph = m.phase_cycles(True); tm = m.timing()
tot = sum(ph); print("timing", {k: round(v,2) for k,v in tm.items() if k.endswith("ms")})
print("phase Mcycles:", [round(p/1e6) for p in ph[:24]])
