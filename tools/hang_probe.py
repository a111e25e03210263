"""debug aid: which call hangs on a 13-template gallery whose last template is empty (faulthandler dumps the stack after 30 s)"""
import faulthandler, importlib, os, sys
faulthandler.dump_traceback_later(30, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
lats = S.make_latents(5, 2); gal = S.make_packed_gallery(5, 12, cb)
m = M.Matcher(cbb, lib_path=sys.argv[1]) if len(sys.argv) > 1 else M.Matcher(cbb)
for j in range(12): m.gallery_add_dat(T.write_rolled(gal.template(j)))
m.gallery_add_dat(b"")
m.gallery_commit(0)
print("search Q=2", flush=True); r = m.search(lats, k=5)
print("search Q=1", flush=True); r = m.search(lats[:1], k=24); print(r["topk_idx"], flush=True)
for idx in ([0, 1], [12], list(range(13))):
    print("corr", idx, flush=True); c = m.correspondences(lats[0], idx); print([None if x[0] is None else len(x[0]) for x in c], flush=True)
print("done")
