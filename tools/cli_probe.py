"""debug aid: runs the match CLI on a tiny synthetic gallery with a timeout and prints what it said"""
import importlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read(); cb = T.Codebook.from_bytes(cbb)
import numpy as np
rng = np.random.default_rng(3)
lats = S.make_latents(5, 2)
tmp = tempfile.mkdtemp()
for d in ("gal", "lat", "o1", "work"): os.mkdir(os.path.join(tmp, d))
gal = S.make_packed_gallery(5, 12, cb)
for j in range(12): open(os.path.join(tmp, "gal", f"R{j:03d}.dat"), "wb").write(T.write_rolled(gal.template(j)))
for i, L in enumerate(lats): open(os.path.join(tmp, "lat", f"L{i}.dat"), "wb").write(T.write_latent(L))
open(os.path.join(tmp, "cb.dat"), "wb").write(cbb)
exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
open(os.path.join(tmp, "gal", "R_empty.dat"), "wb").write(b"")
box = os.path.join(tmp, "cli.afisgal")
def run(args):
    try:
        o = subprocess.run([exe] + args, capture_output=True, text=True, cwd=os.path.join(tmp, "work"), timeout=40)
        print(args[0], "rc", o.returncode, o.stdout[-200:], o.stderr[-300:])
    except subprocess.TimeoutExpired as e:
        print(args[0], "TIMEOUT", (e.stdout or b"")[-400:], (e.stderr or b"")[-400:])
run(["-g", os.path.join(tmp, "gal"), "-pack", box, "-s", os.path.join(tmp, "o1") + "/", "-c", os.path.join(tmp, "cb.dat")])
for mode in (["-ldir", os.path.join(tmp, "lat")], ["-l", os.path.join(tmp, "lat", "L0.dat")]):
    run(mode + ["-g", box, "-s", os.path.join(tmp, "o1") + "/", "-c", os.path.join(tmp, "cb.dat")])
for mode in (["-ldir", os.path.join(tmp, "lat")], ["-l", os.path.join(tmp, "lat", "L0.dat")]):
    try:
        o = subprocess.run([exe] + mode + ["-g", os.path.join(tmp, "gal"), "-s", os.path.join(tmp, "o1") + "/", "-c", os.path.join(tmp, "cb.dat")], capture_output=True, text=True, cwd=os.path.join(tmp, "work"), timeout=40)
        print(mode[0], "rc", o.returncode, o.stdout[-300:], o.stderr[-300:])
    except subprocess.TimeoutExpired as e:
        print(mode[0], "TIMEOUT", (e.stdout or b"")[-400:], (e.stderr or b"")[-400:])
