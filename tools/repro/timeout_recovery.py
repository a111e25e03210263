"""The deadline path of a search, step by step with prints (the body of tests/test_gpu_fullsize.py::test_a_search_that_outlasts_its_deadline...), for a box where it misbehaves.
usage: python tools/repro/timeout_recovery.py [G] [Q] [timeout_ms]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000; Q = int(sys.argv[2]) if len(sys.argv) > 2 else 100; tmo = int(sys.argv[3]) if len(sys.argv) > 3 else 150
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
lats = S.make_latents(909, Q); gal = S.make_packed_gallery(909, G, cb); S.plant_mates(909, gal, cb, lats, G=G)
EARLY = os.environ.get("EARLY_CLOSE") is not None          # stop right after the call that should time out and close the context (what a failing test's teardown does)
m = M.Matcher(cbb, taps=os.environ.get("TAPS") is not None); m.gallery_add_packed(gal); m.gallery_commit(0)
P = lambda *a: print(*a, flush=True)
res = m.search(lats, k=24); P("reference search done", m.timing()["total_ms"])
qh = m.upload_queries(lats); m.search_resident(qh, k=0); P("warm search done, bound_cus", m.get_option("bound_cus"))
m.set_option("search_timeout_ms", tmo)
t0 = time.time()
try:
    m.search_resident(qh, k=0); P("NO timeout?!", time.time() - t0)
except M.AfisError as e:
    P("timed out after %.3f s:" % (time.time() - t0), str(e)[:200])
P("bound_cus now", m.get_option("bound_cus"))
if EARLY:
    P("closing early"); t0 = time.time(); m.close(); P("closed in %.2f s" % (time.time() - t0)); sys.exit(0)
time.sleep(6.0)
m.set_option("search_timeout_s", 600)
P("searching again (one stream)")
t0 = time.time()
try:
    r = m.search([lats[5], lats[50 % Q]], k=24); P("second search done in %.2f s, equal:" % (time.time() - t0), bool(np.array_equal(r["scores"], res["scores"][[5, 50 % Q]])))
except M.AfisError as e:
    P("second search failed after %.2f s:" % (time.time() - t0), str(e)[:300])
m.set_option("bound_cus", 128); P("bound_cus restored", m.get_option("bound_cus"))
r = m.search([lats[5]], k=24); P("third search (overlapped) done, equal:", bool(np.array_equal(r["scores"], res["scores"][[5]])))
m.free_queries(qh); P("closing"); m.close(); P("closed")
