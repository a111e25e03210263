"""Per-phase cycle shares of the tail kernels.  Needs a library built with -DAFIS_PHASE_TIMING:
     cd msu-latentafis_amd/csrc && for f in graph minu; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
        -DAFIS_PHASE_TIMING -c $f.hip -o /tmp/${f}_ph.o; done && hipcc --offload-arch=gfx950 -shared -fPIC adc.o /tmp/graph_ph.o \
        /tmp/minu_ph.o pq_encode.o afis_api.o template_io.o -o ../../tools/libafis_phase.so
   Caveat (graph kernels): the counters are global atomics; a global load that follows them waits for them, so the angle
   stage's share is inflated.  Use the shares inside the distance stage and inside the candidate kernel."""
import sys, importlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
G, Q = 10000, 4
m = M.Matcher(cbb, lib_path=sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "libafis_phase.so"))
if len(sys.argv) > 2 and sys.argv[2].startswith("structured"):                       # python tools/phase_probe.py <lib> structured[:dup]
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured"); sg = SS.DUP_SIGMA[int(sys.argv[2].split(":")[1]) if ":" in sys.argv[2] else 10]
    lats = SS.make_structured_latents(1, Q, sigma=sg); gal = SS.make_packed_gallery_structured(1, G, cb, sigma=sg, encode=m.pq_encode); SS.plant_structured_mates(1, gal, cb, lats, sigma=sg)
else:
    lats = S.make_latents(1, Q); gal = S.make_packed_gallery(1, G, cb); S.plant_mates(1, gal, cb, lats)
m.gallery_add_packed(gal); m.gallery_commit(0); qh = m.upload_queries(lats)
m.search_resident(qh); m.phase_cycles(True)
m.search_resident(qh); ph = m.phase_cycles(True); tm = m.timing()
print("workload", sys.argv[2] if len(sys.argv) > 2 else "headline", "pairs", Q * G)
print("timing", {k: round(v, 2) for k, v in tm.items() if k.endswith("ms")})
names = {16: "fast: load+gemm", 17: "fast: sums/norm", 18: "fast: stage1", 19: "fast: barrier", 20: "fast: stage2 (wave 0)",
         5: "minu graph: dist H bits", 6: "minu graph: dist power iters", 7: "minu graph: dist sort", 8: "minu graph: dist greedy+compact",
         9: "minu graph: angle H bits (inflated)", 10: "minu graph: angle iters", 11: "minu graph: angle sort+greedy",
         21: "tex graph: dist H bits", 22: "tex graph: dist power iters", 23: "tex graph: dist sort", 24: "tex graph: dist greedy+compact",
         25: "tex graph: angle H bits (inflated)", 26: "tex graph: angle iters", 27: "tex graph: angle sort+greedy"}
names[12] = "minu graph: list load"; names[28] = "tex graph: S7 top-200 + list build"
names[29] = "fast:   pass 1: approximate keys + histogram"; names[30] = "fast:   scan for the threshold bin"; names[31] = "fast:   candidate append"
names[16] = "fast: load+gemm (incl. barrier)"; names[18] = "fast:   exact keys of the candidates"; names[20] = "fast: rank + write"
for grp, idxs in (("minutiae candidates", (16, 17, 29, 30, 31, 18, 19, 20)),  ("minutiae graph", range(5, 13)), ("texture graph", range(21, 29))):
    tot = sum(ph[i] for i in idxs) or 1
    print(grp, "total Mcycles", round(tot / 1e6), " = kcycles per list:", round(tot / 1e3 / (Q * G * (3 if "minutiae" in grp else 1)), 2))
    for i in idxs:
        print("   %-36s %6.1f %%" % (names[i], 100.0 * ph[i] / tot))
