"""End-to-end `match` at BASELINE scale on the GPU box (Missing #2): a 100k-template gallery as ONE packed container and as a 100k-file
directory, 100 latents; `match -ldir` (List2List_matching, matcher.cpp:96-214) and `match -l` (One2List_matching, :216-337) timed by the process's own
stage clock (AFIS_MATCH_TIMING: scan / load+parse / commit+upload / latents / search / write) and by wall time, next to the reference-faithful CPU leg
(bench.py's cpu_baseline: the oracle's restatement of the reference loop, every rolled .dat re-parsed per pair) measured on a bounded sample.
  python tools/cli_scale_r04.py [G] [Q] [workdir]      -> one JSON document on stdout (copy it to profiles/r04_cli_scale.json)"""
import importlib, json, os, shutil, subprocess, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 100
work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/afis_cli_scale"
seed = 2024
cbp = os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat")
cbb = open(cbp, "rb").read(); cb = T.Codebook.from_bytes(cbb)
exe = os.path.join(ROOT, "msu-latentafis_amd", "csrc", "match")
shutil.rmtree(work, ignore_errors=True)
for d in ("gal", "lat", "out", "run"): os.makedirs(os.path.join(work, d))
doc = {"G": G, "Q": Q, "seed": seed, "disk_free_GB_before": round(shutil.disk_usage(work).free / 1e9, 1)}

t0 = time.time()
lats = S.make_latents(seed, Q); gal = S.make_packed_gallery(seed, G, cb); planted = S.plant_mates(seed, gal, cb, lats, G=G)
doc["generate_s"] = round(time.time() - t0, 1)
names = [os.path.join(work, "gal", "R%06d.dat" % g) for g in range(G)]

def write_range(lo_hi):
    lo, hi = lo_hi
    for g in range(lo, hi):
        with open(names[g], "wb") as f: f.write(T.write_rolled(gal.template(g)))
    return hi - lo
t0 = time.time()
step = (G + 15) // 16
with ProcessPoolExecutor(16) as ex: n_written = sum(ex.map(write_range, [(a, min(G, a + step)) for a in range(0, G, step)]))
doc["write_gallery_files_s"] = round(time.time() - t0, 1); doc["gallery_dir_GB"] = round(sum(os.path.getsize(n) for n in names[:2000]) / 2000 * G / 1e9, 2)
for i, L in enumerate(lats):
    with open(os.path.join(work, "lat", "L%03d.dat" % i), "wb") as f: f.write(T.write_latent(L))
t0 = time.time()
m = M.Matcher(cbb); m.gallery_add_packed(gal); box = os.path.join(work, "gallery.afisgal"); m.gallery_save(box, names); m.close()
doc["write_container_s"] = round(time.time() - t0, 1); doc["container_GB"] = round(os.path.getsize(box) / 1e9, 2)

def run(tag, args):
    out = os.path.join(work, "out", tag) + "/"; os.makedirs(out)
    t = time.time()
    r = subprocess.run([exe, *args, "-s", out, "-c", cbp, "-d", "0"], cwd=os.path.join(work, "run"), env=dict(os.environ, AFIS_MATCH_TIMING="1"), stdout=open(os.path.join(work, tag + ".stdout"), "w"), stderr=subprocess.PIPE, text=True)
    wall = time.time() - t
    line = [l for l in r.stderr.splitlines() if "timing (ms)" in l]
    stages = {}
    if line:
        w = line[-1].split("timing (ms):")[1].split()
        stages = {w[i]: float(w[i + 1]) for i in range(0, len(w), 2)}
    dline = [l for l in r.stderr.splitlines() if "on the device's clock (ms):" in l]
    if dline:
        w = dline[-1].split("(ms):")[1].split()
        stages.update({"device_" + w[i]: float(w[i + 1]) for i in range(0, len(w), 2)})
    files = os.listdir(out)
    total = [l for l in open(os.path.join(work, tag + ".stdout")) if l.startswith("Total matching duration")]
    return {"rc": r.returncode, "wall_s": round(wall, 2), "stages_ms": stages, "reported_total_ms": float(total[-1].split(":")[1]) if total else None,
            "commit_laps": [l.strip() for l in r.stderr.splitlines() if l.startswith("commit:")],          # only with AFIS_COMMIT_TIMING=1 in the environment
            "files_written": len(files), "bytes_written": sum(os.path.getsize(os.path.join(out, f)) for f in files), "stderr_tail": r.stderr[-300:] if r.returncode else ""}, out

runs = {}
runs["l_container"], o1 = run("l_container", ["-l", os.path.join(work, "lat", "L000.dat"), "-g", box])
runs["ldir_container"], o2 = run("ldir_container", ["-ldir", os.path.join(work, "lat"), "-g", box])
runs["l_directory"], o3 = run("l_directory", ["-l", os.path.join(work, "lat", "L000.dat"), "-g", os.path.join(work, "gal")])
runs["ldir_directory"], o4 = run("ldir_directory", ["-ldir", os.path.join(work, "lat"), "-g", os.path.join(work, "gal")])
doc["runs"] = runs
# sanity: the planted true mate of latent 0 leads its rank list; the -ldir files have G lines; container and directory give the same scores for a latent
first = open(os.path.join(o1, "L000.csv")).read().splitlines()[1]
doc["l_rank1_is_planted_mate"] = ("R%06d.dat" % planted[0][0][0]) in first
a = open(os.path.join(o2, "L000.csv")).read().splitlines()
doc["ldir_lines_per_file"] = len(a)
sc_box = np.array([float(l.rsplit(",", 1)[1]) for l in a])
b = {l.rsplit(",", 1)[0]: float(l.rsplit(",", 1)[1]) for l in open(os.path.join(o4, "L000.csv")).read().splitlines()}
doc["container_equals_directory_scores"] = bool(all(b[l.rsplit(",", 1)[0]] == float(l.rsplit(",", 1)[1]) for l in a))
doc["ldir_top_score_latent0"] = float(sc_box.max())
# the reference-faithful CPU leg on a bounded sample (bench.py's own definition), for the ratio
sys.argv = [sys.argv[0]]
bench = importlib.import_module("bench")
cpu = bench.cpu_baseline(cbb, lats, gal, 0, pairs_per_thread=400)
doc["cpu_reference_faithful_pairs_per_s_8_threads"] = round(cpu["pairs_per_s_reference_faithful"], 1)
doc["cpu_compute_only_best_pairs_per_s"] = cpu["pairs_per_s"]; doc["cpu_threads_best"] = cpu["threads"]; doc["cpu_limits"] = cpu["limits"]
doc["cpu_reference_faithful_extrapolated_s_for_this_job"] = round(Q * G / cpu["pairs_per_s_reference_faithful"], 0)
if not os.environ.get("AFIS_CLI_KEEP"): shutil.rmtree(work, ignore_errors=True)       # AFIS_CLI_KEEP=1: leave the gallery for tools/ldir_repeat.sh
print(json.dumps(doc, indent=1))
