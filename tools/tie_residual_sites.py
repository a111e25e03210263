#!/usr/bin/env python
"""With std::sort's order reproduced at S3 (option s3_tie_order 1 = oracle tie mode 4), which sites hold what is LEFT against the reference binary's order (tie mode 0)?
CPU only: the oracle's modes 6/7/8 take std::sort at S3 plus ONE more site (S7 / S8 / S9).  90 000 structured pairs (the named workload, dup 10), about 13 minutes on 64 threads.
Result (profiles/r06_tie_residual_sites.txt): 13 pairs beyond the tolerance with S3 alone; 12 with S3 + S7, 7 with S3 + S8, 7 with S3 + S9 — the rest is split between the greedy
selections of S8 and S9 (six pairs each), S7 holds one.  Modes 9 (S3 + S8 + S9 = option ref_tie_order 2) and 1 (the default) and the PLANTED MATES are listed too: the
S8 / S9 pairs are the mates (93 of 120 differ in a bit under the default order, none under mode 9: profiles/r06_parity_sweep_ref_tie_order2.txt)."""
import importlib, os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from oracle_lib import Oracle
T = importlib.import_module("msu-latentafis_amd.host.templates"); SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
cbb = open(ROOT+"/tests/golden/codebook_EmbeddingSize_96_stride_16_subdim_6.dat","rb").read(); cb = T.Codebook.from_bytes(cbb)
seed, Q, G = 2073, 30, 3000
sg = SS.DUP_SIGMA[10]
lats = SS.make_structured_latents(seed, Q, sigma=sg); gal = SS.make_packed_gallery_structured(seed, G, cb, sigma=sg, encode=cb.encode_fast); planted = SS.plant_structured_mates(seed, gal, cb, lats, G=G, sigma=sg)
orc = Oracle(); ocb = orc.codebook(cbb); nt = orc.lib.orc_num_threads()
hr = [orc.rolled(T.write_rolled(gal.template(g)))[0] for g in range(G)]
modes=(9,4,6,7,8,3,5,1)
far={m:0 for m in modes}; bit={m:0 for m in modes}; partfar={m:np.zeros(4,int) for m in modes}; mate_bit={m:0 for m in modes}; mate_far={m:0 for m in modes}; n_mates=0
t0=time.time()
for qi,L in enumerate(lats):
    hl,_=orc.latent(ocb, T.write_latent(L))
    rc,s0,p0=orc.search(ocb,hl,hr,tie_mode=0,threads=nt,want_parts=True)
    mates=[g for g,_f in planted[qi]]; n_mates+=len(mates)
    for m in modes:
        rc,s1,p1=orc.search(ocb,hl,hr,tie_mode=m,threads=nt,want_parts=True)
        f=np.abs(s0-s1)>1e-3*np.maximum(1,np.abs(s0)); far[m]+=int(f.sum()); bit[m]+=int((s0.view(np.uint32)!=s1.view(np.uint32)).sum())
        mate_bit[m]+=int((s0.view(np.uint32)!=s1.view(np.uint32))[mates].sum()); mate_far[m]+=int(f[mates].sum())
        partfar[m]+= (np.abs(p0[:,:4]-p1[:,:4])>1e-3*np.maximum(1,np.abs(p0[:,:4]))).sum(axis=0)
    orc.lib.orc_latent_free(hl)
print("pairs",Q*G,"time",time.time()-t0)
print("planted mates", n_mates)
for m in modes: print(m, "beyond 1e-3:",far[m],"bit:",bit[m],"parts(minu x3, tex):",partfar[m],"| planted mates with a differing bit:",mate_bit[m],"beyond 1e-3:",mate_far[m])
