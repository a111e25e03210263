import sys, importlib, os
sys.path.insert(0, "/root/repo")
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open("/root/repo/tests/golden/codebook_EmbeddingSize_96_stride_16_subdim_6.dat","rb").read(); cb = T.Codebook.from_bytes(cbb)
G, Q = 10000, 4
lats = S.make_latents(1, Q); gal = S.make_packed_gallery(1, G, cb)
for name in ["libafis_hip.so"] + [f"libafis_ab{i}.so" for i in (4, 5)]:
    path = os.path.join(os.path.dirname(M.LIB_PATH), name)
    if not os.path.exists(path): continue
    m = M.Matcher(cbb, lib_path=path); m.gallery_add_packed(gal); m.gallery_commit(0); qh = m.upload_queries(lats)
    for v in (4, 5):
        m.set_option("adc_variant", v); m.search_resident(qh); m.search_resident(qh); print(name, "variant", v, "adc_ms %.2f" % m.timing()["adc_ms"])
    m.close()
