import importlib, sys, time, numpy as np, os
sys.path.insert(0,'/root/repo')
T = importlib.import_module("msu-latentafis_amd.host.templates"); SS = importlib.import_module("msu-latentafis_amd.host.synth_structured"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb=open('/root/repo/tests/golden/codebook_EmbeddingSize_96_stride_16_subdim_6.dat','rb').read(); cb=T.Codebook.from_bytes(cbb)
def sets(kind):
    rng = np.random.default_rng(905)
    if kind == "structured":
        SS.IDENTITY_WEIGHT = 1.0
        lats = [SS.make_structured_latent(rng, sigma=0.0095, n_tex_lo=200, n_tex_hi=260) for _ in range(2)]
        gal = [SS.make_structured_rolled(rng, cb, sigma=0.0095, n_minu=int(rng.integers(20, 128)), n_tex=300) for _ in range(60)]
    else:
        lats = [S.make_latent(rng, n_tex_lo=210, n_tex_hi=256) for _ in range(2)]
        gal = [S.make_rolled(rng, cb, n_tex=300) for _ in range(60)]
    return lats, gal
for lib in (sys.argv[1:] or ["cur"]):
    for kind in ("iid", "structured"):
        lats, gal = sets(kind)
        m = M.Matcher(cbb, lib_path=None if lib == "cur" else lib); m.gallery_add(gal); m.gallery_commit(0)
        for qi in (0, 1, 0):
            t=time.time(); r=m.search([lats[qi]], k=0); dt=time.time()-t; tm=m.timing()
            print(lib[-20:], kind, "rows", lats[qi].tex[0].n, "search", round(dt,2), "tex_tail_ms", round(tm["tex_tail_ms"],1), flush=True)
        m.close()
