"""CPU tests of the host logic: .dat / codebook formats (Python mirror and the C++ implementation the CLI and C ABI use), the
C-ABI surface, and the synthetic generator."""
import ctypes
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

import cases

T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")
M = importlib.import_module("msu-latentafis_amd.host.matcher")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "msu-latentafis_amd", "csrc")


def fnv(parts):
    h = 1469598103934665603
    for b in parts:
        for byte in b:
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def cb(codebook_bytes):
    return T.Codebook.from_bytes(codebook_bytes)


@pytest.fixture(scope="module")
def tio():
    exe = os.path.join(CSRC, "tio_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", CSRC, "tio_check"], check=True)
    return exe


def test_codebook_shipped_file(codebook_bytes, cb, tio, tmp_path):
    assert len(codebook_bytes) == 98310 and (cb.M, cb.K, cb.dsub) == (16, 256, 6)
    assert cb.to_bytes() == codebook_bytes
    assert -0.65 < cb.words.min() < -0.64 and 0.51 < cb.words.max() < 0.52
    p = tmp_path / "cb.dat"; p.write_bytes(codebook_bytes)
    out = subprocess.run([tio, "codebook", str(p)], capture_output=True, text=True, check=True).stdout
    assert "ok=1 M=16 K=256 dsub=6" in out
    assert int(re.search(r"hash=([0-9a-f]+)", out).group(1), 16) == fnv([cb.words.tobytes()])
    with pytest.raises(ValueError):
        T.Codebook.from_bytes(b"\x10\x00\x00\x01\x06\x00")


def test_pq_encode_picks_nearest_codeword(cb):
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 256, (50, 16))
    des = np.stack([np.concatenate([cb.words[m, idx[i, m]] for m in range(16)]) for i in range(50)]).astype(np.float32)
    assert np.array_equal(cb.encode(des), idx.astype(np.uint8))
    assert np.array_equal(cb.encode(des + rng.normal(0, 1e-4, des.shape).astype(np.float32)), idx.astype(np.uint8))


def test_dat_roundtrip_python_and_cpp(cb, tio, tmp_path):
    lats, gal = cases.small_set(cb, seed=2, n_lat=1, n_gal=4)
    lb, rb = T.write_latent(lats[0]), T.write_rolled(gal[0])
    rc, L = T.read_latent(lb)
    assert rc == 0 and len(L.minu) == 28 and len(L.tex) == 1 and np.array_equal(L.tex[0].des, lats[0].tex[0].des)
    assert np.array_equal(L.minu[26].des, lats[0].minu[26].des) and np.array_equal(L.minu[2].x, lats[0].minu[2].x)
    rc, R = T.read_rolled(rb)
    assert rc == 0 and np.array_equal(R.tex[0].codes, gal[0].tex[0].codes) and np.array_equal(R.minu[0].ori, gal[0].minu[0].ori)
    assert T.write_latent(L) == lb and T.write_rolled(R) == rb
    # header layout of descriptor_PQ.py:87-107: 12 x u16 (version 1 first), h, w, blkH, blkW, u8 count
    assert lb[:2] == b"\x01\x00" and lb[2:24] == b"\x00" * 22 and lb[32] == 28 and rb[32] == 1
    for kind, buf, t in (("latent", lb, L), ("rolled", rb, R)):
        p = tmp_path / f"{kind}.dat"; p.write_bytes(buf)
        out = subprocess.run([tio, kind, str(p)], capture_output=True, text=True, check=True).stdout.splitlines()
        assert out[0].startswith(f"rc=0 n_minu={len(t.minu)} n_tex={len(t.tex)} ")
        hashes = [int(re.search(r"hash=([0-9a-f]+)", l).group(1), 16) for l in out[1:]]
        want = [fnv([m.x.astype("<i2").tobytes(), m.y.astype("<i2").tobytes(), m.ori.tobytes(), m.des.tobytes()]) for m in t.minu]
        want += [fnv([x.x.astype("<i2").tobytes(), x.y.astype("<i2").tobytes(), x.ori.tobytes(),
                      x.des.tobytes() if x.des is not None else b"", x.codes.tobytes() if x.codes is not None else b""]) for x in t.tex]
        assert hashes == want
        rt = subprocess.run([tio, f"roundtrip-{kind}", str(p)], capture_output=True, text=True, check=True).stdout
        assert "identical=1" in rt


def test_reader_edge_cases(cb, tio, tmp_path):
    # empty / tiny files (matcher.cpp:798-801, :899-902)
    assert T.read_latent(b"")[0] == 1 and T.read_rolled(b"\x00" * 10)[0] == 1
    lats, gal = cases.small_set(cb, seed=3, n_lat=1, n_gal=2)
    # more than 2000 texture points -> -1 and the rolled template is treated as empty (matcher.cpp:865-869, :173-177)
    big = T.FPTemplate(minu=list(gal[0].minu), tex=list(gal[0].tex))
    buf = bytearray(T.write_rolled(big))
    off = 33 + 2 + gal[0].minu[0].n * (2 + 2 + 4) + 2 + gal[0].minu[0].n * 96 * 4 + 1
    buf[off:off + 2] = (2001).to_bytes(2, "little")
    rc, t = T.read_rolled(bytes(buf))
    assert rc == -1 and t.minu == [] and t.tex == []
    p = tmp_path / "big.dat"; p.write_bytes(bytes(buf))
    assert subprocess.run([tio, "rolled", str(p)], capture_output=True, text=True).stdout.startswith("rc=-1")
    # truncated file: what was read before the end is kept, nothing after (ifstream semantics)
    rb = T.write_rolled(gal[1])
    rc, t = T.read_rolled(rb[:len(rb) - 100])
    assert rc == 0 and len(t.tex) == 1 and np.array_equal(t.tex[0].codes[:-7], gal[1].tex[0].codes[:-7]) and not t.tex[0].codes[-1].any()
    # texture counts above 1000 are legal on disk (clamped at match time, matcher.cpp:544-547)
    assert T.MAX_NROF_MINUTIAE == 2000
    # a descriptor length outside 1..192 (the reference overruns a stack buffer there): both parsers stop with code 8 and parse NOTHING
    # from the misaligned bytes that follow; what came before is kept
    ok = bytearray(T.write_rolled(gal[0]))
    dl_off = 33 + 2 + gal[0].minu[0].n * (2 + 2 + 4)
    assert int.from_bytes(ok[dl_off:dl_off + 2], "little") == 96
    bad = bytearray(ok); bad[dl_off:dl_off + 2] = (300).to_bytes(2, "little")
    rc, t = T.read_rolled(bytes(bad))
    assert rc == T.BAD_DES_LENGTH == 8 and t.minu == [] and t.tex == []
    p = tmp_path / "baddl.dat"; p.write_bytes(bytes(bad))
    out = subprocess.run([tio, "rolled", str(p)], capture_output=True, text=True).stdout
    assert out.startswith("rc=8") and "n_minu=0 n_tex=0" in out.splitlines()[0], out
    tex_dl = off + 2 + gal[0].tex[0].n * (2 + 2 + 4)
    assert int.from_bytes(ok[tex_dl:tex_dl + 2], "little") == 16
    bad = bytearray(ok); bad[tex_dl:tex_dl + 2] = (0).to_bytes(2, "little")
    rc, t = T.read_rolled(bytes(bad))
    assert rc == 8 and len(t.minu) == 1 and t.tex == []
    # a gallery container only takes 96-float descriptors / 16-byte codes: a legal-but-different width is refused, not over-read
    narrow = T.FPTemplate(minu=[T.MinutiaeTemplate(gal[0].minu[0].x, gal[0].minu[0].y, gal[0].minu[0].ori, gal[0].minu[0].des[:, :12].copy())], tex=[])
    p = tmp_path / "narrow.dat"; p.write_bytes(T.write_rolled(narrow))
    r = subprocess.run([tio, "gallery-pack", str(tmp_path / "x.afisgal"), str(p)], capture_output=True, text=True)
    assert r.returncode == 3 and "descriptor width" in r.stderr


def test_header_only_template_when_no_minutiae(cb):
    """descriptor_PQ.py:90-93: a template without minutiae templates is 12 shorts + 4 zero shorts and nothing else."""
    t = T.FPTemplate()
    assert T.write_rolled(t) == b"\x01\x00" + b"\x00" * 22 + b"\x00" * 8 and T.write_latent(t) == T.write_rolled(t)
    rc, r = T.read_rolled(T.write_rolled(t))
    assert rc == 0 and r.minu == [] and r.tex == []


def test_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "afis_matcher.h")).read()
    declared = sorted(set(re.findall(r"\b(afis_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    lib = M.load_library()                       # dlopen only: no device call
    for name in declared:
        assert hasattr(lib, name), name
    assert set(M.EXPORTS) == set(declared)
    assert not any(n.startswith("afis_debug") for n in declared)          # the product ABI carries no parity taps ...
    for name in M.TAP_EXPORTS:
        assert not hasattr(lib, name), name                                # ... and the product library exports none
    # no torch / C++ types in the signatures
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # signatures only, comments stripped
    assert "std::" not in code and "torch" not in code and "hip" not in code.lower() and "&" not in code
    # the test library = the product's exports + exactly the taps of include/afis_matcher_taps.h
    taps_hdr = open(os.path.join(ROOT, "include", "afis_matcher_taps.h")).read()
    tap_decl = sorted(set(re.findall(r"\b(afis_debug_[a-z0-9_]+)\s*\(", taps_hdr)))
    assert set(M.TAP_EXPORTS) == set(tap_decl)
    tlib = M.load_library(M.TEST_LIB_PATH)
    for name in declared + tap_decl:
        assert hasattr(tlib, name), name


def test_cli_help_runs_without_gpu():
    exe = os.path.join(CSRC, "match")
    if not os.path.exists(exe):
        pytest.skip("match not built")
    out = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "-ldir" in out.stdout and "-c <codebook.dat>" in out.stdout


def test_synthetic_gallery_is_shard_consistent(cb):
    G = 2500
    full = S.make_packed_gallery(9, G, cb)
    part = S.make_packed_gallery(9, G, cb, 700, 2100)
    ref = full.slice(700, 2100)
    for f in ("minu_off", "minu_x", "minu_des", "tex_off", "tex_codes", "tex_ori"):
        assert np.array_equal(getattr(part, f), getattr(ref, f)), f
    lats = S.make_latents(9, 3, n_tex_lo=220, n_tex_hi=260)
    p1 = S.plant_mates(9, full, cb, lats)
    p2 = S.plant_mates(9, part, cb, lats, G=G, lo=700)
    assert p1 == p2
    ref = full.slice(700, 2100)
    for f in ("minu_x", "minu_des", "tex_codes", "tex_x"):
        assert np.array_equal(getattr(part, f), getattr(ref, f)), f
    nm, nt = S.gallery_counts(9, G)
    assert nm.min() >= 20 and nm.max() <= 200 and nt.min() >= 600 and nt.max() <= 1000
    assert np.allclose(np.linalg.norm(full.minu_des[:100], axis=1), 1.73, atol=1e-4)


# ---- packed gallery container: C++ writer/reader (tio_check, host only) against the Python mirror -------------
def _small_gallery(cb, n=7, seed=3):
    rng = np.random.default_rng(seed)
    ts = [S.make_rolled(rng, cb, n_tex=int(rng.integers(20, 60)), n_minu=int(rng.integers(5, 15))) for _ in range(n)]
    ts[2] = T.FPTemplate(minu=ts[2].minu, tex=[])            # no texture template
    ts[4] = T.FPTemplate()                                   # empty file
    big = S.make_rolled(rng, cb, n_tex=1100, n_minu=8)       # above the 1000-point clamp (matcher.cpp:546-547)
    ts.append(big)
    return ts


def _pack(ts):
    mo, to = [0], [0]; mx, my, mori, mdes, tx, ty, tori, tc = [], [], [], [], [], [], [], []
    for t in ts:
        if t.minu:
            m = t.minu[0]; mx.append(m.x); my.append(m.y); mori.append(m.ori); mdes.append(m.des)
        mo.append(mo[-1] + (t.minu[0].n if t.minu else 0))
        n = min(t.tex[0].n, 1000) if t.tex else 0
        if n:
            x = t.tex[0]; tx.append(x.x[:n]); ty.append(x.y[:n]); tori.append(x.ori[:n]); tc.append(x.codes[:n])
        to.append(to[-1] + n)
    cat = lambda a, dt, shape=(0,): np.concatenate(a).astype(dt) if a else np.zeros(shape, dt)
    return S.PackedGallery(np.array(mo, np.int64), cat(mx, np.int16), cat(my, np.int16), cat(mori, np.float32), cat(mdes, np.float32, (0, 96)),
                           np.array(to, np.int64), cat(tx, np.int16), cat(ty, np.int16), cat(tori, np.float32), cat(tc, np.uint8, (0, 16)))


def _dump_hashes(out):
    return {k: v for k, v in re.findall(r"(\w+) hash=([0-9a-f]+)", out)}


def _want_hashes(g, names, tex_counts):
    empty = ((np.diff(g.minu_off) == 0) & (np.diff(g.tex_off) == 0)).astype(np.uint8)
    return {"offsets": "%016x" % fnv([g.minu_off.astype("<i8").tobytes(), g.tex_off.astype("<i8").tobytes(), empty.tobytes()]),
            "minutiae": "%016x" % fnv([g.minu_x.tobytes(), g.minu_y.tobytes(), g.minu_ori.tobytes(), g.minu_des.tobytes()]),
            "texture": "%016x" % fnv([g.tex_x.tobytes(), g.tex_y.tobytes(), g.tex_ori.tobytes(), g.tex_codes.tobytes()]),
            "names": "%016x" % fnv([n.encode() + b"\0" for n in names]),
            "tex_counts": "%016x" % fnv([np.asarray(tex_counts, "<i4").tobytes()])}


def test_gallery_container_cpp_and_python_agree(cb, tio, tmp_path):
    CT = importlib.import_module("msu-latentafis_amd.host.container")
    ts = _small_gallery(cb)
    files = []
    for i, t in enumerate(ts):
        p = tmp_path / f"R{i:02d}.dat"; p.write_bytes(T.write_rolled(t)); files.append(str(p))
    g = _pack(ts)
    tex_counts = np.diff(g.tex_off)
    # C++ packs the .dat files; Python reads the container back
    cpp = tmp_path / "cpp.afisgal"
    out = subprocess.run([tio, "gallery-pack", str(cpp)] + files, capture_output=True, text=True, check=True).stdout
    assert out.strip() == f"ok G={len(ts)}"
    g2, names2, tc2 = CT.read_container(str(cpp))
    for f in ("minu_off", "minu_x", "minu_y", "minu_ori", "minu_des", "tex_off", "tex_x", "tex_y", "tex_ori", "tex_codes"):
        assert np.array_equal(getattr(g2, f), getattr(g, f)), f
    assert names2 == files and np.array_equal(tc2, tex_counts) and tc2.max() == 1000
    # Python writes; C++ reads (whole file and a shard range) and reports hashes of what it loaded
    py = tmp_path / "py.afisgal"
    CT.write_container(str(py), g, files)
    assert py.read_bytes() == cpp.read_bytes()                         # the two writers produce the same file
    out = subprocess.run([tio, "gallery-dump", str(py)], capture_output=True, text=True, check=True).stdout
    assert f"G={len(ts)} n_minu={len(g.minu_x)} n_tex={len(g.tex_x)} range={len(ts)}" in out
    assert _dump_hashes(out) == _want_hashes(g, files, tex_counts)
    first, count = 2, 4
    out = subprocess.run([tio, "gallery-dump", str(py), str(first), str(count)], capture_output=True, text=True, check=True).stdout
    sub, sub_names, _ = CT.read_container(str(py), first, count)
    assert np.array_equal(sub.minu_des, _pack(ts[first:first + count]).minu_des)
    assert _dump_hashes(out) == _want_hashes(sub, sub_names, tex_counts)
    # the container read in place (what afis_gallery_load keeps mapped and afis_gallery_commit uploads from): the same arrays, whole and as a shard
    for args, gg in (([], g), ([str(first), str(count)], sub)):
        out = subprocess.run([tio, "gallery-map", str(py)] + args, capture_output=True, text=True, check=True).stdout
        want = _want_hashes(gg, [], [])
        assert _dump_hashes(out) == {k: want[k] for k in ("offsets", "minutiae", "texture")}, out
    assert "error=" in subprocess.run([tio, "gallery-map", str(py), "5", "9"], capture_output=True, text=True).stdout
    # damaged files are rejected, not read past their end
    bad = tmp_path / "bad.afisgal"; bad.write_bytes(py.read_bytes()[:-100])
    assert "error=" in subprocess.run([tio, "gallery-dump", str(bad)], capture_output=True, text=True).stdout
    assert "error=" in subprocess.run([tio, "gallery-map", str(bad)], capture_output=True, text=True).stdout
    bad.write_bytes(b"NOTAGAL1" + py.read_bytes()[8:])
    assert "error=" in subprocess.run([tio, "gallery-dump", str(bad)], capture_output=True, text=True).stdout
    assert "error=" in subprocess.run([tio, "gallery-map", str(bad)], capture_output=True, text=True).stdout
    out = subprocess.run([tio, "gallery-dump", str(py), "5", "9"], capture_output=True, text=True).stdout
    assert "error=" in out and "range" in out
    with pytest.raises(ValueError):
        CT.read_container(str(bad))


def test_readers_survive_damaged_files(cb, tio, tmp_path):
    """Truncated and bit-flipped templates / containers: the C++ readers return a code or an error, they never crash or hang
    (the reference's ifstream-based reader likewise just stops filling its buffers at EOF, matcher.cpp:785-983)."""
    rng = np.random.default_rng(17)
    lat = T.write_latent(S.make_latent(rng, n_tex_lo=60, n_tex_hi=80, n_minu_lo=5, n_minu_hi=9))
    rol = T.write_rolled(S.make_rolled(rng, cb, n_tex=70, n_minu=12))
    cases_ = []
    for kind, buf in (("latent", lat), ("rolled", rol)):
        for cut in [0, 1, 5, 10, 11, 23, 24, 25, 33, 34, 35, 40, len(buf) // 3, len(buf) // 2, len(buf) - 1]:
            cases_.append((kind, buf[:cut]))
        for _ in range(40):
            b = bytearray(buf)
            for pos in rng.integers(0, min(len(b), 400), 3):          # headers and counts are where damage matters
                b[pos] = int(rng.integers(0, 256))
            cases_.append((kind, bytes(b)))
    for i, (kind, data) in enumerate(cases_):
        p = tmp_path / f"d{i}.dat"; p.write_bytes(data)
        r = subprocess.run([tio, kind, str(p)], capture_output=True, text=True, timeout=20)
        assert r.returncode in (0, 1) and r.stdout.startswith("rc="), (kind, i, r.returncode, r.stderr[:200])
        # the Python mirror agrees on the return code and the template counts
        rc, t = (T.read_latent if kind == "latent" else T.read_rolled)(data)
        m = re.match(r"rc=(-?\d+) n_minu=(\d+) n_tex=(\d+)", r.stdout)
        assert m and int(m.group(1)) == rc and (int(m.group(2)), int(m.group(3))) == (len(t.minu), len(t.tex)) or rc < 0, (kind, i, r.stdout[:80], rc)
    CT = importlib.import_module("msu-latentafis_amd.host.container")
    good = tmp_path / "g.afisgal"
    files = []
    for i in range(3):
        f = tmp_path / f"R{i}.dat"; f.write_bytes(T.write_rolled(S.make_rolled(rng, cb, n_tex=30, n_minu=6))); files.append(str(f))
    subprocess.run([tio, "gallery-pack", str(good)] + files, check=True, capture_output=True)
    raw = good.read_bytes()
    for i in range(60):
        b = bytearray(raw)
        if i % 3 == 0:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            for pos in rng.integers(8, 160, 2):
                b[pos] = int(rng.integers(0, 256))
        p = tmp_path / f"c{i}.afisgal"; p.write_bytes(bytes(b))
        for mode in ("gallery-dump", "gallery-map"):                  # copied into host arrays / read in place: both reject or read, neither crashes
            r = subprocess.run([tio, mode, str(p)], capture_output=True, text=True, timeout=20)
            assert r.returncode in (0, 1) and (r.stdout.startswith("G=") or r.stdout.startswith("error=")), (mode, i, r.returncode, r.stdout[:100], r.stderr[:200])


def test_device_atan2f_restatement_equals_libm_on_the_host(tmp_path):
    """csrc/atan2f_libm.h (the angle stage's atan2, compiled here for the host) against the C library's atan2f on every integer
    coordinate difference |d| <= 700 and a non-integer sweep; the GPU test repeats it on the device over |d| <= 2047."""
    exe = tmp_path / "atan2f_check"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-o", str(exe), os.path.join(ROOT, "tools", "atan2f_check.c"), "-lm"], check=True)
    out = subprocess.run([str(exe), "700"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "1962801 points, 0 mismatches" in out.stdout and "16000000 points, 0 mismatches" in out.stdout


def test_candidate_shape_class_rule_keeps_every_task_inside_its_class_limits():
    """The rule that routes a minutiae candidate task (nL latent x nR rolled minutiae) to a shape class of k_minu_cands_rt (afis_device.h: rt_max_rows) is host-side arithmetic the
    kernels TRUST: a task admitted to class S must fit the class's similarity matrix in LDS with its row stride, its rows must be covered by the keys a thread holds (32, the large
    class twice that) times the row phases 256 S / nR, the stride must be odd (conflict-free sums; the one exception is the small class's 128), and a larger class must take at least
    what a smaller one takes.  Checked for every nR the reader accepts (matcher.cpp:788-790: up to 2000)."""
    import subprocess
    csrc = os.path.join(ROOT, "msu-latentafis_amd", "csrc")
    exe = os.path.join(csrc, "match_selftest")
    subprocess.run(["make", "-s", "-C", csrc, "match_selftest"], check=True)
    out = subprocess.run([exe, "-selftest-classes"], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    simi1, simi2, simi4, k1, k2, k4, max_r, max_l = (int(x) for x in out[0].split())
    simi = {1: simi1, 2: simi2, 4: simi4}; keys = {1: k1, 2: k2, 4: k4}
    assert (simi1, k1, k2, k4, max_r, max_l) == (8192, 32, 32, 64, 512, 256)
    assert 4 * simi4 + 7192 <= 160 * 1024                      # the large class's matrix + its other LDS arrays fit a CU's 160 KB
    rows = [[int(x) for x in l.split()] for l in out[1:]]
    assert [r[0] for r in rows] == list(range(2001))
    for nR, L1, L2, L4, s1, s2, s4 in rows:
        L = {1: L1, 2: L2, 4: L4}; st = {1: s1, 2: s2, 4: s4}
        assert 0 <= L1 <= L2 <= L4 <= 256, nR
        if nR == 0 or nR > 512: assert L4 == 0, nR
        for S_ in (1, 2, 4):
            if L[S_] == 0: continue
            assert st[S_] >= nR and (st[S_] % 2 == 1 or (S_ == 1 and nR == 128)), (S_, nR)
            assert L[S_] * st[S_] <= simi[S_], (S_, nR)                                   # the matrix fits
            assert L[S_] <= keys[S_] * ((256 * S_) // nR), (S_, nR)                         # every row has a key slot
            assert (256 * S_) // nR >= 2, (S_, nR)                                          # at least two row phases: the selection's thread mapping
            max_rolled = {1: 128, 2: 256, 4: 512}[S_]
            assert nR <= max_rolled and max_rolled + L[S_] <= 256 * S_, (S_, nR)               # thread j sums column j, thread max_rolled + i sums row i: both exist
    assert rows[128][1] == 64 and rows[129][1] == 0 and rows[128][2] == 128 and rows[129][2] == 96 and rows[256][2] == 64 and rows[256][3] == 151 and rows[257][2] == 0 and rows[400][3] == 97 and rows[512][3] == 75 and rows[513][3] == 0


def test_launch_group_rule_places_the_cuts_where_the_row_groups_are_fewest(tmp_path):
    """The latents of a search are cut into launch groups on the host (afis_device.h: launch_group_cuts / launch_group_latents; used by afis_queries_upload).  With the matrix-core
    bound pass a launch pays for row groups of 768 latent texture rows, so the rule must (1) cover the latents with contiguous runs of at most `per`, (2) reach the smallest total
    number of row groups any such partition has — checked against an independent dynamic programme here —, (3) among those use the fewest launches; the plain form is runs of `per`.
    The bench's 100 latents (seed 2024) become 2 + 49 + 49 at 50 per launch: 88 row groups, what ONE launch of all of them would need.  The automatic group size is about five
    million pairs per launch, between 10 and 128 latents."""
    import subprocess
    import importlib
    S = importlib.import_module("msu-latentafis_amd.host.synth")
    csrc = os.path.join(ROOT, "msu-latentafis_amd", "csrc")
    exe = os.path.join(csrc, "match_selftest")
    subprocess.run(["make", "-s", "-C", csrc, "match_selftest"], check=True)

    def run(rows, per, plain=False):
        f = tmp_path / "rows.txt"
        f.write_text("".join("%d\n" % r for r in rows))
        out = subprocess.run([exe, "-selftest-groups", str(f), "-per", str(per)] + (["-plain"] if plain else []), capture_output=True, text=True, check=True).stdout.splitlines()
        return [int(x) for x in out[0].split()], dict((int(a), int(b)) for a, b in (t.split(":") for t in out[1].split()))

    def cost(rows, cuts):
        c, prev = 0, 0
        for e in cuts:
            c += (sum(rows[prev:e]) + 767) // 768; prev = e
        return c

    def optimum(rows, per):                                              # (row groups, launches) of the best partition, independently
        n = len(rows); pre = np.concatenate([[0], np.cumsum(rows)])
        best = [(0, 0)] + [(1 << 60, 0)] * n
        for i in range(1, n + 1):
            best[i] = min((best[j][0] + (int(pre[i] - pre[j]) + 767) // 768, best[j][1] + 1) for j in range(max(0, i - per), i))
        return best[n]

    rng = np.random.default_rng(5)
    cases = [([700], 50), ([0, 0, 0], 2), ([768] * 7, 3), ([1000] * 130, 128), ([1] * 40, 7)]
    for _ in range(40):
        n = int(rng.integers(1, 60))
        cases.append((list(int(x) for x in rng.integers(0, 1001, n)), int(rng.integers(1, 20))))
    for rows, per in cases:
        cuts, _ = run(rows, per)
        assert cuts == sorted(set(cuts)) and cuts[-1] == len(rows) and all(b - a <= per for a, b in zip([0] + cuts, cuts)), (rows, per, cuts)
        assert (cost(rows, cuts), len(cuts)) == optimum(rows, per), (rows, per, cuts)
        plain, _ = run(rows, per, plain=True)
        assert plain == list(range(per, len(rows), per)) + [len(rows)]
    lats = S.make_latents(2024, 100, **S.WORKLOADS["headline"]["latent"])
    rows = [min(L.tex[0].n, 1000) for L in lats]
    cuts, auto = run(rows, 50)
    assert cuts == [2, 51, 100] and cost(rows, cuts) == (sum(rows) + 767) // 768 == 88
    assert auto == {1: 128, 12500: 128, 39000: 128, 40000: 125, 50000: 100, 100000: 50, 250000: 20, 500000: 10, 1000000: 10}


def test_product_library_reads_only_the_operational_variables():
    """INTEGRATION.md section E lists the environment variables the product library reads ("Operational"); experiment knobs (result-changing or known-bad ones among
    them) live in the test library only.  `strings libafis_hip.so | grep ^AFIS_` must be exactly the library's part of that table."""
    def env_names(path):
        data = open(path, "rb").read()
        return set(m.decode() for m in re.findall(rb"(?<![A-Z_])AFIS_[A-Z0-9_]+(?=\x00)", data))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("### Environment variables"):doc.index("That table is the whole list")]
    listed = set(re.findall(r"`(AFIS_[A-Z0-9_]+)`", "\n".join(row.split("|")[1] for row in table.splitlines() if row.startswith("| `AFIS_"))))   # first column of the table's rows
    not_the_library = {"AFIS_EXCHANGE", "AFIS_EXCHANGE_TIMEOUT_S", "AFIS_FORCE_EXCHANGE", "AFIS_MATCH_TIMING"}      # read by `match` / libafis_exchange.so (same table, their own rows)
    assert env_names(os.path.join(CSRC, "libafis_hip.so")) == listed - not_the_library
    experiments = {"AFIS_MF_NO_XCD_MAP", "AFIS_BOUND_WHOLE_XCDS", "AFIS_GROUP_WAIT", "AFIS_WAIT_CTX_SYNC_ONLY", "AFIS_ABLATE_SKIP_TEXTURE_TAIL"}
    assert env_names(os.path.join(CSRC, "libafis_hip_test.so")) == (listed - not_the_library) | experiments
    for name in experiments:
        assert name in doc[doc.index("That table is the whole list"):doc.index("## F.")], name


def test_structured_generator_has_the_structure_it_claims(cb):
    """host/synth_structured.py: unique grid coordinates in scan order (extraction_rolled.py:112-128), orientation = minus a smooth flow direction (:125), codes from a
    smooth field (neighbours share most codes; the named duplicate shares), latents with two orientations per grid point (extraction_latent.py:204-205), and shard
    consistency (a shard's content depends on (seed, template index) only)."""
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
    G = 1300
    full = SS.make_packed_gallery_structured(5, G, cb, sigma=SS.DUP_SIGMA[10], n_tex_lo=300, n_tex_hi=500)
    part = SS.make_packed_gallery_structured(5, G, cb, 900, 1200, sigma=SS.DUP_SIGMA[10], n_tex_lo=300, n_tex_hi=500)
    a, b = int(full.tex_off[900]), int(full.tex_off[1200])
    assert np.array_equal(full.tex_codes[a:b], part.tex_codes) and np.array_equal(full.tex_x[a:b], part.tex_x) and np.array_equal(full.tex_ori[a:b], part.tex_ori)
    ma, mb = int(full.minu_off[900]), int(full.minu_off[1200])
    assert np.array_equal(full.minu_des[ma:mb], part.minu_des) and np.array_equal(full.minu_x[ma:mb], part.minu_x)
    nm, nt = S.gallery_counts(5, G, n_tex_lo=300, n_tex_hi=500)
    assert np.array_equal(np.diff(full.tex_off), nt) and np.array_equal(np.diff(full.minu_off), nm)
    n_smooth = n_nb = 0
    for g in range(0, 200):
        t = full.template(g).tex[0]
        cell = t.y.astype(np.int64) * SS.BLK_W + t.x
        assert np.all(np.diff(cell) > 0)                                            # unique, y-major / x-minor
        assert t.x.min() >= 0 and t.x.max() < SS.BLK_W and t.y.max() < SS.BLK_H
        assert np.all(np.abs(t.ori) <= np.pi / 2 + 1e-6)
        nb = (t.y[1:] == t.y[:-1]) & (t.x[1:] == t.x[:-1] + 1)
        d = np.abs(t.ori[1:] - t.ori[:-1])[nb]; d = np.minimum(d, np.pi - d)        # orientation mod pi
        n_smooth += int((d < 0.35).sum()); n_nb += int(nb.sum())
    assert n_smooth > 0.97 * n_nb                                                   # smooth but for the few blocks next to a singular point
    share = {d: SS.dup_share(*(lambda gl: (gl.tex_codes, gl.tex_off))(SS.make_packed_gallery_structured(6, 64, cb, sigma=SS.DUP_SIGMA[d]))) for d in (0, 10, 30)}
    assert share[0] < 0.01 and 0.05 < share[10] < 0.16 and 0.22 < share[30] < 0.40, share
    few = 0                                                                         # (latent, rolled) minutiae pairs that keep fewer than 120 positive similarities: rare at the named identity weight
    Lm = SS.make_structured_latent(np.random.default_rng(8)).minu[26]
    for g in range(300):
        a_, b_ = int(full.minu_off[g]), int(full.minu_off[g + 1])
        few += int(((Lm.des @ full.minu_des[a_:b_].T) > 0).sum() < 120)
    assert few <= 6, few
    same = (full.tex_y[1:] == full.tex_y[:-1]) & (full.tex_x[1:] == full.tex_x[:-1] + 1)
    assert (full.tex_codes[1:] == full.tex_codes[:-1]).sum(1)[same].mean() > 9      # of 16 codes, with the right-hand neighbour
    L = SS.make_structured_latent(np.random.default_rng(3))
    lt = L.tex[0]
    assert lt.n % 2 == 0 and np.array_equal(lt.x[0::2], lt.x[1::2]) and np.array_equal(lt.y[0::2], lt.y[1::2])
    assert np.allclose(lt.ori[1::2] - lt.ori[0::2], np.pi, atol=1e-6)
    assert len(L.minu) == 28 and np.allclose(np.linalg.norm(lt.des, axis=1), T.DESCRIPTOR_NORM, atol=1e-4)
    R = SS.make_structured_mate(np.random.default_rng(4), cb, L, frac=0.8)
    rc = R.tex[0].y.astype(np.int64) * SS.BLK_W + R.tex[0].x
    assert len(np.unique(rc)) == len(rc)                                            # a mate keeps unique coordinates
    rc2, back = T.read_rolled(T.write_rolled(R))
    assert rc2 == 0 and np.array_equal(back.tex[0].codes, R.tex[0].codes)


def test_stdsort_order_equals_libstdcxx(tmp_path):
    """csrc/stdsort_order.h — the restatement of libstdc++'s std::sort that option s3_tie_order 1 runs on the device to order equal candidate norms as the reference binary does
    (matcher.cpp:473-476) — against std::sort itself on the host: 40 000 arrays (distinct keys, few distinct values, mostly-zero keys as in a list short of 120 positive
    similarities, all equal, organ pipes; every n up to 300 and task-shaped ones up to 8192), the first K positions only (the pruned form the kernel uses) and whole arrays,
    and forced depth limits so that the heap-sort branch runs."""
    exe = tmp_path / "stdsort_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "stdsort_check.cpp")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout + out.stderr


def test_rank_list_entry_point_is_the_reference_sort(oracle):
    """afis_rank_list (include/afis_matcher.h; host only: no device needed): matcher.cpp:306-309's rank list from a score column.  ref_order 1 = std::sort on the reference's
    comparator — the oracle's orc_rank_list calls the same libstdc++ routine —, ref_order 0 = equal scores by ascending index; k beyond n pads with -1."""
    import ctypes as C
    lib = M.load_library()
    rng = np.random.default_rng(11)
    for n in (0, 1, 16, 17, 66, 5000):
        s = np.zeros(n, np.float32)
        if n: s[rng.choice(n, max(1, n // 5), replace=False)] = (rng.random(max(1, n // 5)) * 50).astype(np.float32)
        if n > 20: s[3] = s[7] = 12.5                                                  # a tie among positive scores as well
        for ref in (0, 1):
            k = 24
            idx = np.full(k, 99, np.int64); sc = np.full(k, 9.0, np.float32)
            rc = lib.afis_rank_list(s.ctypes.data_as(C.POINTER(C.c_float)) if n else None, C.c_int64(n), C.c_int(ref), C.c_int(k), idx.ctypes.data_as(C.POINTER(C.c_int64)), sc.ctypes.data_as(C.POINTER(C.c_float)))
            assert rc == 0
            want = oracle.rank_list(s, bool(ref))[:k] if n else np.zeros(0, np.int32)
            m = min(k, n)
            assert np.array_equal(idx[:m], want[:m]) and np.array_equal(sc[:m], s[want[:m]]) and (idx[m:] == -1).all() and (sc[m:] == 0).all(), (n, ref)
    assert lib.afis_rank_list(None, C.c_int64(5), C.c_int(0), C.c_int(3), None, None) != 0

