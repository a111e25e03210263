"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Tolerances: S4-S7 bit-exact; fused scores |d| <= 1e-3*max(1,|s|) for >= 99.9% of pairs, rank lists identical
over strictly positive non-tied scores.  The oracle runs in tie_mode=1 (equal keys by ascending index), the order the
HIP path implements.
"""
import importlib
import os

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")
M = importlib.import_module("msu-latentafis_amd.host.matcher")
SH = importlib.import_module("msu-latentafis_amd.host.sharding")


@pytest.fixture(scope="module")
def cb(codebook_bytes):
    return T.Codebook.from_bytes(codebook_bytes)


@pytest.fixture(scope="module")
def small(cb):
    return cases.small_set(cb)


def _matcher(codebook_bytes, gal, variant=None, taps=False):
    taps = taps or (variant is not None and variant < 8)   # the direct exact kernels (adc_variant 0-3, 6, 7: the second witness of the row maxima) are built into the test library only
    m = M.Matcher(codebook_bytes, taps=taps)        # taps: libafis_hip_test.so (the product objects + afis_debug_* + the reference kernels); default = the product library
    if variant is not None:
        m.set_option("adc_variant", variant)
    m.gallery_add(gal)
    m.gallery_commit(0)
    return m


def test_lut_bit_exact(codebook_bytes, cb, oracle, small):
    lats, gal = small
    m = _matcher(codebook_bytes, gal[:2], taps=True)
    ocb = oracle.codebook(codebook_bytes)
    for L in lats:
        got = m.debug_lut(L)
        want = oracle.build_lut(ocb, L.tex[0].des)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 6, 7, 8, 9])
def test_rowmax_bit_exact(codebook_bytes, cb, oracle, small, variant):
    lats, gal = small
    m = _matcher(codebook_bytes, gal, variant, taps=True)
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)
    for qi in range(len(lats)):
        for g in (0, 1, 2, 5, 17, len(gal) - 1):
            val, arg = m.debug_texture_rowmax(lats[qi], g)
            oval, oarg = oracle.texture_rowmax(ocb, hl[qi], hr[g])
            assert np.array_equal(val.view(np.uint32), oval.view(np.uint32)), (variant, qi, g)
            assert np.array_equal(arg, oarg), (variant, qi, g)


def _compare(res, orc_parts, tol=1e-3):
    got = np.concatenate([res["parts"], res["scores"][..., None]], axis=-1)
    want = orc_parts
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    return got, want, err


@pytest.mark.parametrize("variant", [0, 1, 7, 9])
def test_scores_small(codebook_bytes, cb, oracle, small, variant):
    lats, gal = small
    m = _matcher(codebook_bytes, gal, variant)
    res = m.search(lats, k=5, want_parts=True)
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)
    want = np.zeros((len(lats), len(gal), 5), np.float32)
    for qi, h in enumerate(hl):
        rc, sc, parts = oracle.search(ocb, h, hr, tie_mode=1, want_parts=True)
        assert rc == 0
        want[qi] = parts
    got, want, err = _compare(res, want)
    assert (want[..., 4] > 0).sum() >= 3 * len(lats) - 2, "planted mates must score"
    bad = got.view(np.uint32) != want.view(np.uint32)               # DESIGN section 2 claims 0 ulp against tie_mode 1: assert exactly that
    assert not bad.any(), f"pairs with a differing bit: {np.argwhere(bad)[:10]}, got {got[bad][:5]}, want {want[bad][:5]}"
    # rank lists: the planted mates come first, in the oracle's order
    for qi in range(len(lats)):
        order = np.lexsort((np.arange(len(gal)), -want[qi, :, 4]))[:5]
        pos = want[qi, order, 4] > 0
        assert np.array_equal(res["topk_idx"][qi][pos], order[pos])


def test_correspondences_match_oracle(codebook_bytes, cb, oracle, small):
    """Correspondence export (matcher.cpp:497-505): survivors of both graph filters, as coordinates, in the oracle's order."""
    lats, gal = small
    m = _matcher(codebook_bytes, gal)
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)
    sel = (26, 2, 11)
    n_nonempty = 0
    for qi, L in enumerate(lats):
        got = m.correspondences(L, list(range(len(gal))))
        for gi, R in enumerate(gal):
            for s in range(3):
                tr = oracle.trace(ocb, hl[qi], hr[gi], which=s + 1, stage=2, tie_mode=1)
                if tr is None or not R.minu:
                    assert got[gi][s] is None
                    continue
                _, li, ri = tr
                lm, rm = L.minu[sel[s]], R.minu[0]
                want = np.stack([lm.x[li], lm.y[li], rm.x[ri], rm.y[ri]], axis=1).astype(np.int16) if len(li) else np.zeros((0, 4), np.int16)
                assert np.array_equal(got[gi][s], want), (qi, gi, s)
                n_nonempty += len(li) > 0
    assert n_nonempty >= 3, "planted mates must have surviving correspondences"
    # out-of-shard index is an error, not a silent zero
    with pytest.raises(M.AfisError):
        m.correspondences(lats[0], [len(gal)])
    m.close()


def test_pq_encoder_bit_exact(codebook_bytes, cb, oracle, tmp_path):
    """GPU nearest-codeword encoder vs the scipy golden, the oracle, and through the file / gallery entry points."""
    import os, subprocess
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_pq.npz"))
    m = M.Matcher(codebook_bytes)
    assert np.array_equal(m.pq_encode(g["des"]), g["codes"])                      # scipy.cluster.vq.vq's answers
    rng = np.random.default_rng(8)
    ocb = oracle.codebook(codebook_bytes)
    for n in (1, 63, 64, 65, 5000):                                                 # tile remainders
        des = rng.standard_normal((n, 96)).astype(np.float32) * rng.choice([0.05, 0.1, 0.3], (n, 1)).astype(np.float32)
        des[::7, :6] = cb.words[0][3]                                               # exact hits; duplicated codewords would tie -> first
        assert np.array_equal(m.pq_encode(des), oracle.pq_encode(ocb, des)), n
    assert m.pq_encode(np.zeros((0, 96), np.float32)).shape == (0, 16)
    # file entry point: latent-layout template (fp32 texture descriptors) -> rolled-layout template with codes
    src = S.make_latent(rng, n_tex_lo=300, n_tex_hi=340)
    src = T.FPTemplate(minu=src.minu[:1], tex=src.tex[:1])
    rc, out = m.encode_rolled_dat(T.write_latent(src))
    assert rc == 0
    rrc, R = T.read_rolled(out)
    assert rrc == 0 and len(R.minu) == 1 and len(R.tex) == 1
    assert np.array_equal(R.tex[0].codes, oracle.pq_encode(ocb, src.tex[0].des))
    assert np.array_equal(R.tex[0].x, src.tex[0].x) and np.array_equal(R.tex[0].ori, src.tex[0].ori)
    assert np.array_equal(R.minu[0].des, src.minu[0].des)
    assert oracle.rolled(out)[1] == 0                                               # the oracle's reader accepts the file
    # gallery entry point: fp32 texture descriptors are encoded at afis_gallery_add; same scores as the pre-encoded template
    lat = S.make_latent(rng)
    m2 = M.Matcher(codebook_bytes); m2.gallery_add([R]); m2.gallery_commit(0)
    m3 = M.Matcher(codebook_bytes); m3.gallery_add([T.FPTemplate(minu=src.minu[:1], tex=src.tex[:1])]); m3.gallery_commit(0)
    assert np.array_equal(m2.search([lat], k=0)["scores"], m3.search([lat], k=0)["scores"])
    m2.close(); m3.close(); m.close()
    # command line (descriptor_PQ.py's arguments): directory in, directory out, digit order, 2-byte file for a template without texture
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "pq_encode")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(M.LIB_PATH), "pq_encode"], check=True)
    (tmp_path / "in").mkdir(); (tmp_path / "work").mkdir()
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    (tmp_path / "in" / "r10.dat").write_bytes(T.write_latent(src))
    (tmp_path / "in" / "r9.dat").write_bytes(T.write_latent(T.FPTemplate(minu=src.minu[:1], tex=[])))
    out = subprocess.run([exe, "--fprint_type", "rolled", "--input_dir", str(tmp_path / "in") + "/", "--output_dir", str(tmp_path / "enc") + "/", "-c", str(cbp)],
                         capture_output=True, text=True, cwd=tmp_path / "work")
    assert out.returncode == 0, out.stderr
    pq_lines = [l for l in out.stdout.splitlines() if l.startswith("PQ: ")]
    assert [os.path.basename(l) for l in pq_lines] == ["r9.dat", "r10.dat"]         # sorted by the digits, not lexicographically
    assert (tmp_path / "enc" / "r9.dat").read_bytes() == b"\x00\x00"
    rrc2, R2 = T.read_rolled((tmp_path / "enc" / "r10.dat").read_bytes())
    assert rrc2 == 0 and np.array_equal(R2.tex[0].codes, R.tex[0].codes)
    out = subprocess.run([exe, "--fprint_type", "rolled", "--input_file", "x.dat", "--output_dir", str(tmp_path / "enc") + "/"], capture_output=True, text=True, cwd=tmp_path / "work")
    assert "Single template PQ is not available for rolled prints" in out.stdout
    out = subprocess.run([exe, "--fprint_type", "rolled"], capture_output=True, text=True, cwd=tmp_path / "work")
    assert "Missing args." in out.stdout


def test_gallery_container_equals_directory(codebook_bytes, cb, small, tmp_path):
    """a gallery loaded from one packed container (whole, and as two shards) scores exactly like the same gallery
    loaded from its .dat files; the CLI prints the same score files from either."""
    import os, subprocess
    lats, gal = small
    (tmp_path / "gal").mkdir(); (tmp_path / "lat").mkdir(); (tmp_path / "o1").mkdir(); (tmp_path / "o2").mkdir(); (tmp_path / "work").mkdir()
    files = []
    for j, g in enumerate(gal[:12]):
        p = tmp_path / "gal" / f"R{j:03d}.dat"; p.write_bytes(T.write_rolled(g)); files.append(str(p))
    p = tmp_path / "gal" / "R_empty.dat"; p.write_bytes(b""); files.append(str(p))
    m = M.Matcher(codebook_bytes)
    m.gallery_reserve(len(files))                                   # a hint; changes nothing that can be observed
    for i, f in enumerate(files):
        m.gallery_add_dat(open(f, "rb").read())
        if i == 2: m.gallery_reserve(len(files))
    box = str(tmp_path / "g.afisgal")
    m.gallery_save(box, files)
    m.gallery_commit(0)
    with pytest.raises(M.AfisError):
        m.gallery_save(box, files)                                  # the staging copy is gone after commit
    want = m.search(lats[:2], k=5)
    G, nm, nt, tc = m.gallery_file_info(box)
    assert G == 13 and nt == tc.sum() and tc[-1] == 0 and m.gallery_file_names(box, 3, 2) == files[3:5]
    m1 = M.Matcher(codebook_bytes); m1.gallery_load(box); m1.gallery_commit(0)
    got = m1.search(lats[:2], k=5)
    assert np.array_equal(got["scores"], want["scores"]) and np.array_equal(got["topk_idx"], want["topk_idx"])
    bounds = SH.shard_bounds(tc, 2)                                 # two shards straight from the file, balanced by texture points
    parts = []
    for lo, hi in bounds:
        ms = M.Matcher(codebook_bytes); ms.gallery_load(box, lo, hi - lo); ms.gallery_commit(lo)
        parts.append(ms.search(lats[:2], k=5)); ms.close()
    assert np.array_equal(np.concatenate([p["scores"] for p in parts], axis=1), want["scores"])
    with pytest.raises(M.AfisError):
        m1.gallery_load(box)                                        # committed
    # a load into an empty staging area only MAPS the file (the commit uploads from the mapping); anything staged after it first copies the mapped range
    # into host arrays.  Both routes, and two loads in a row, give the gallery the files give.
    m3 = M.Matcher(codebook_bytes); m3.gallery_load(box, 0, 5)
    assert m3.gallery_size == 5
    m3.gallery_load(box, 5, 4)
    for f in files[9:]:
        m3.gallery_add_dat(open(f, "rb").read())
    assert m3.gallery_size == 13
    box3 = str(tmp_path / "again.afisgal"); m3.gallery_save(box3, files)
    assert open(box3, "rb").read() == open(box, "rb").read()
    m3.gallery_commit(0)
    got = m3.search(lats[:2], k=5)
    assert np.array_equal(got["scores"], want["scores"]) and np.array_equal(got["topk_idx"], want["topk_idx"])
    m3.close()
    m4 = M.Matcher(codebook_bytes); m4.gallery_load(box, 12, 1); assert m4.gallery_size == 1; m4.gallery_commit(12)      # a shard of one empty template, straight from the mapping
    got = m4.search(lats[:2], k=1)
    assert np.array_equal(got["scores"], want["scores"][:, 12:13]); m4.close()
    m4 = M.Matcher(codebook_bytes); m4.gallery_load(box, 3, 0); assert m4.gallery_size == 0; m4.gallery_commit(3); m4.close()   # an empty shard
    m2 = M.Matcher(codebook_bytes)
    with pytest.raises(M.AfisError):
        m2.gallery_load(box, 10, 9)                                 # range outside the file
    m.close(); m1.close(); m2.close()
    # CLI: -pack writes the container, -g <container> gives the same outputs as -g <directory>
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    for i, L in enumerate(lats[:2]):
        (tmp_path / "lat" / f"L{i}.dat").write_bytes(T.write_latent(L))
    box2 = str(tmp_path / "cli.afisgal")
    out = subprocess.run([exe, "-g", str(tmp_path / "gal"), "-pack", box2, "-s", str(tmp_path / "o1") + "/", "-c", str(cbp)], capture_output=True, text=True, cwd=tmp_path / "work")
    assert out.returncode == 0 and "Packed 13 templates" in out.stdout, out.stderr
    for mode in (["-ldir", str(tmp_path / "lat")], ["-l", str(tmp_path / "lat" / "L0.dat")]):
        o1 = subprocess.run([exe] + mode + ["-g", str(tmp_path / "gal"), "-s", str(tmp_path / "o1") + "/", "-c", str(cbp)], capture_output=True, text=True, cwd=tmp_path / "work")
        o2 = subprocess.run([exe] + mode + ["-g", box2, "-s", str(tmp_path / "o2") + "/", "-c", str(cbp)], capture_output=True, text=True, cwd=tmp_path / "work")
        assert o1.returncode == 0 and o2.returncode == 0, (o1.stderr, o2.stderr)
        names = sorted(os.listdir(tmp_path / "o1"))
        assert names == sorted(os.listdir(tmp_path / "o2")) and len(names) >= 2
        for n in names:
            assert (tmp_path / "o1" / n).read_bytes() == (tmp_path / "o2" / n).read_bytes(), n


def test_all_templates_mode(codebook_bytes, cb, oracle, small):
    """One2One_matching_all_templates (matcher.cpp:339-374) — every latent minutiae template and every latent
    texture template — against the oracle, incl. a latent with two texture templates, one without texture, and empty prints."""
    lats, gal = small
    rng = np.random.default_rng(77)
    two_tex = T.FPTemplate(minu=lats[0].minu[:7], tex=[lats[0].tex[0], S.make_latent(rng, n_tex_lo=200, n_tex_hi=260).tex[0]])
    no_tex = T.FPTemplate(minu=lats[1].minu[:4], tex=[])
    g = list(gal[:6]) + [T.FPTemplate(), T.FPTemplate(minu=gal[0].minu, tex=[]), T.FPTemplate(minu=[], tex=gal[1].tex)]
    m = M.Matcher(codebook_bytes)
    for r in g:                                          # through the file format, as the oracle (a rolled file without minutiae
        m.gallery_add_dat(T.write_rolled(r))             # templates carries no texture either, descriptor_PQ.py:190-193)
    m.gallery_commit(0)
    ocb = oracle.codebook(codebook_bytes)
    hr = [oracle.rolled(T.write_rolled(r))[0] for r in g]
    for L in (lats[0], two_tex, no_tex):
        qs, rs, sc = m.One2One_matching_all_templates(L)
        hl, _ = oracle.latent(ocb, T.write_latent(L))
        width = len(L.minu) + len(L.tex)
        assert qs == 0 and sc.shape == (len(g), width)
        n_pos = 0
        for gi in range(len(g)):
            rc, want = oracle.all_templates(ocb, hl, hr[gi], width, tie_mode=1)
            assert rs[gi] == (2 if rc == 2 else 0)
            err = np.abs(sc[gi] - want) / np.maximum(1, np.abs(want))
            assert (err <= 1e-3).all(), (gi, sc[gi], want)
            assert np.array_equal(sc[gi].view(np.uint32), want.view(np.uint32))
            n_pos += int((want > 0).sum())
        assert n_pos > 0
    qs, rs, sc = m.One2One_matching_all_templates(T.FPTemplate())
    assert qs == 1 and sc.shape == (len(g), 0)
    # the selected-template scores are columns 26, 2, 11 and n_minu of the all-template vector
    r = m.search([lats[0]], k=0, want_parts=True)
    qs, rs, sc = m.One2One_matching_all_templates(lats[0])
    p = r["parts"].reshape(len(g), 4)
    ok = rs == 0
    assert np.array_equal(p[ok][:, :3], sc[ok][:, [26, 2, 11]]) and np.array_equal(p[ok][:, 3], sc[ok][:, len(lats[0].minu)])
    m.close()


def test_stage_lists_match_oracle_traces(codebook_bytes, cb, oracle, small):
    """Every scorer's correspondence list after every stage — candidates (S3 / S7), distance filter (S8), angle filter (S9) —
    against the oracle's traces: same members, same order, same raw similarities (bit for bit).  Final scores alone would not
    notice a 1-ulp slip inside the graph stages; this does as soon as a selection changes."""
    lats, gal = small
    m = _matcher(codebook_bytes, gal, taps=True)
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)
    n_lists = n_nonempty_final = 0
    for qi in range(len(lats)):
        for gi in range(len(gal)):
            for which in range(4):
                for stage in range(3):
                    want = oracle.trace(ocb, hl[qi], hr[gi], which=which, stage=stage, tie_mode=1)
                    got = m.debug_stage_list(lats[qi], gi, which, stage)
                    if want is None:
                        assert got is None, (qi, gi, which, stage)
                        continue
                    assert got is not None, (qi, gi, which, stage)
                    ws, wl, wr = want
                    assert np.array_equal(got[1], wl) and np.array_equal(got[2], wr), (qi, gi, which, stage)
                    assert np.array_equal(got[0].view(np.uint32), ws.view(np.uint32)), (qi, gi, which, stage)
                    n_lists += 1
                    n_nonempty_final += int(stage == 2 and len(wl) > 0)
    assert n_lists >= 100 and n_nonempty_final >= 6
    m.close()


def test_c_abi_error_behaviour(codebook_bytes, cb, small):
    """The boundary fails loudly and leaves the context usable: state errors, bad views, bad options, bad files."""
    import ctypes as C
    lats, gal = small
    m = M.Matcher(codebook_bytes)
    lib = m.lib
    with pytest.raises(M.AfisError, match="commit"):
        m.search(lats[:1])                                           # search before commit
    with pytest.raises(M.AfisError, match="commit"):
        m.correspondences(lats[0], [0])
    m.gallery_add(gal[:3])
    bad = T.FPTemplate(minu=[T.MinutiaeTemplate(gal[0].minu[0].x, gal[0].minu[0].y, gal[0].minu[0].ori, gal[0].minu[0].des[:, :64].copy())], tex=[])
    with pytest.raises(M.AfisError, match="96"):
        m.gallery_add([bad])                                         # the reference asserts equal descriptor lengths (matcher.cpp:433)
    assert m.gallery_size == 3                                       # a rejected template adds nothing
    m.gallery_commit(0)
    with pytest.raises(M.AfisError, match="committed"):
        m.gallery_add(gal[:1])
    with pytest.raises(M.AfisError, match="committed"):
        m.gallery_commit(0)
    with pytest.raises(M.AfisError):
        m.set_option("adc_variant", 4)                               # removed variants
    with pytest.raises(M.AfisError, match="libafis_hip_test"):
        m.set_option("adc_variant", 7)                               # the direct kernels are reference kernels of the test library, not product surface
    with pytest.raises(M.AfisError, match="libafis_hip_test"):
        m.set_option("mf_blocks", 3)
    with pytest.raises(M.AfisError):
        m.set_option("no_such_option", 1)
    with pytest.raises(M.AfisError, match="96"):
        m.search([T.FPTemplate(minu=[bad.minu[0]] * 28, tex=[])])
    with pytest.raises(M.AfisError):
        m.gallery_file_info("/nonexistent/gallery.afisgal")
    assert lib.afis_search(m.ctx, None, 1, None, None, None, 0, None, None) != 0         # NULL queries with n_q > 0
    assert lib.afis_search(m.ctx, None, 0, None, None, None, 0, None, None) == 0         # an empty batch is fine
    assert lib.afis_destroy(None) is None                                                  # no-op
    ctx2 = C.c_void_p()
    assert lib.afis_create_from_codebook(C.byref(ctx2), b"\x00" * 20, 20, 0) != 0 and not ctx2.value      # not a codebook
    assert lib.afis_create_from_codebook(C.byref(ctx2), codebook_bytes, len(codebook_bytes), 99) != 0      # no such device
    assert b"device" in lib.afis_last_error(None)
    r = m.search(lats[:1], k=5)                                      # still works after all of the above; k > G pads with -1
    assert list(r["topk_idx"][0][3:]) == [-1, -1] and (r["topk_idx"][0][:3] >= 0).all()
    m.close()


def test_candidate_selection_degenerate_keys(codebook_bytes, cb, oracle):
    """S3 with keys that defeat the histogram path of k_minu_cands_fast (n >= 512): (a) fewer than 120 non-zero similarities, so
    the 120th candidate is a zero decided by element index; (b) every similarity identical, so all keys share one bin and the
    whole top-120 is decided by index; (c) a block of exact ties straddling the 120th place.  Candidate lists (members, order,
    similarity bits) and scores against the oracle."""
    rng = np.random.default_rng(5)
    base = S.make_latent(rng)
    d = rng.standard_normal(96).astype(np.float32); d *= np.float32(1.73) / np.linalg.norm(d)
    e = rng.standard_normal(96).astype(np.float32); e -= d * (e @ d) / (d @ d); e *= np.float32(1.73) / np.linalg.norm(e)     # orthogonal to d

    def latent_with(des_rows):
        L = T.FPTemplate(minu=list(base.minu), tex=list(base.tex))
        m0 = base.minu[26]
        n = len(des_rows)
        L.minu[26] = T.MinutiaeTemplate(rng.integers(50, 700, n).astype(np.int16), rng.integers(50, 700, n).astype(np.int16),
                                        rng.uniform(-3, 3, n).astype(np.float32), np.stack(des_rows).astype(np.float32))
        return L

    def rolled_with(des_rows):
        n = len(des_rows)
        R = S.make_rolled(rng, cb, n_minu=n, n_tex=300)
        R.minu[0] = T.MinutiaeTemplate(R.minu[0].x, R.minu[0].y, R.minu[0].ori, np.stack(des_rows).astype(np.float32))
        return R

    cases_ = [
        (latent_with([d] * 30), rolled_with([-d] * 37 + [d] * 3)),                      # (a) 90 non-zero of 1200
        (latent_with([d] * 32), rolled_with([d] * 40)),                                  # (b) 1280 identical keys
        (latent_with([d] * 10 + [e] * 22), rolled_with([d] * 10 + [0.5 * d + 0.5 * e] * 30)),   # (c) a few levels, large tie blocks
    ]
    ocb = oracle.codebook(codebook_bytes)
    for ci, (L, R) in enumerate(cases_):
        m = M.Matcher(codebook_bytes, taps=True); m.gallery_add([R]); m.gallery_commit(0)
        hl, _ = oracle.latent(ocb, T.write_latent(L)); hr, _ = oracle.rolled(T.write_rolled(R))
        for stage in (0, 1, 2):
            want = oracle.trace(ocb, hl, hr, which=1, stage=stage, tie_mode=1)
            got = m.debug_stage_list(L, 0, 1, stage)
            assert want is not None and got is not None
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (ci, stage)
            assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (ci, stage)
            if stage == 0:
                assert len(want[1]) == 120
        rc, want_sc = oracle.pair(ocb, hl, hr, 1)
        got_sc = m.search([L], k=0, want_parts=True)
        assert np.array_equal(got_sc["parts"][0, 0].view(np.uint32), want_sc[:4].view(np.uint32)), ci
        m.close()


def test_cli_exchange_path_single_rank(codebook_bytes, cb, small, tmp_path):
    """`match` with the multi-rank code path switched on for one rank (RCCL communicator of size 1, the all-gathers, the merge, the
    shard plan): same files as the plain single-process run, for -l (rank list + correspondence files) and -ldir."""
    import os, subprocess
    lats, gal = small
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
    for d in ("gal", "lat", "o1", "o2", "work"):
        (tmp_path / d).mkdir()
    for j, g in enumerate(gal[:10]):
        (tmp_path / "gal" / f"R{j:03d}.dat").write_bytes(T.write_rolled(g))
    (tmp_path / "gal" / "R_empty.dat").write_bytes(b"")
    for i, L in enumerate(lats[:2]):
        (tmp_path / "lat" / f"L{i}.dat").write_bytes(T.write_latent(L))
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    env = dict(os.environ, AFIS_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    env.pop("NCCL_DEBUG", None)
    for mode in (["-ldir", str(tmp_path / "lat")], ["-l", str(tmp_path / "lat" / "L0.dat")]):
        common = ["-g", str(tmp_path / "gal"), "-c", str(cbp)]
        o1 = subprocess.run([exe] + mode + common + ["-s", str(tmp_path / "o1") + "/"], capture_output=True, text=True, cwd=tmp_path / "work")
        o2 = subprocess.run([exe] + mode + common + ["-s", str(tmp_path / "o2") + "/"], capture_output=True, text=True, cwd=tmp_path / "work", env=env)
        assert o1.returncode == 0 and o2.returncode == 0, (o1.stderr, o2.stderr)
        names = sorted(os.listdir(tmp_path / "o1"))
        assert names == sorted(os.listdir(tmp_path / "o2")) and len(names) >= 2
        for n in names:
            assert (tmp_path / "o1" / n).read_bytes() == (tmp_path / "o2" / n).read_bytes(), n
        strip = lambda t: [l for l in t.splitlines() if not l.startswith("Total matching duration")]
        assert strip(o1.stdout) == strip(o2.stdout)


def test_texture_coordinates_off_the_beaten_path(codebook_bytes, cb, oracle):
    """S8b range rule and arithmetic paths: block coordinates spread over 0..120 (many pairs with |dx| or |dy| >= 50, which the
    table look-up treats as incompatible, matcher.cpp:1257), beyond 8191 (the kernel's packed 16-bit fast path must step aside) and
    >= 32768 (negative as the reference reads them into int16-backed storage).  Stage lists and scores against the oracle."""
    rng = np.random.default_rng(21)
    base = S.make_latent(rng, n_tex_lo=260, n_tex_hi=300)
    ocb = oracle.codebook(codebook_bytes)

    def spread(t, scale, offset=0):
        x = ((t.x.astype(np.int64) * scale) // 10 + offset).astype(np.int64)
        y = ((t.y.astype(np.int64) * scale) // 10 + offset).astype(np.int64)
        return x.astype(np.uint16).view(np.int16), y.astype(np.uint16).view(np.int16)

    for ci, (scale, off_l, off_r) in enumerate(((25, 0, 0), (25, 9000, 9000), (10, 40000, 40000), (25, 0, 8180))):
        L = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(*spread(base.tex[0], scale, off_l), base.tex[0].ori, des=base.tex[0].des)])
        R0 = S.make_mate(rng, cb, base, frac=0.7, n_tex=500)
        rx, ry = spread(R0.tex[0], scale, off_r)
        R = T.FPTemplate(minu=list(R0.minu), tex=[T.TextureTemplate(rx, ry, R0.tex[0].ori, codes=R0.tex[0].codes)])
        m = M.Matcher(codebook_bytes, taps=True); m.gallery_add_dat(T.write_rolled(R)); m.gallery_commit(0)
        hl, _ = oracle.latent(ocb, T.write_latent(L)); hr, _ = oracle.rolled(T.write_rolled(R))
        for stage in (0, 1, 2):
            want = oracle.trace(ocb, hl, hr, which=0, stage=stage, tie_mode=1)
            got = m.debug_stage_list(L, 0, 0, stage)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (ci, stage)
            assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (ci, stage)
        rc, want_sc = oracle.pair(ocb, hl, hr, 1)
        got_sc = m.search([L], k=0, want_parts=True)["parts"][0, 0]
        assert np.array_equal(got_sc.view(np.uint32), want_sc[:4].view(np.uint32)), (ci, got_sc, want_sc)
        m.close()


def test_texture_top200_with_tied_row_maxima(codebook_bytes, cb, oracle):
    """S7 (matcher.cpp:736-749) when row maxima tie: latent texture rows that share a descriptor have bit-identical ADC rows, hence equal
    maxima and arg-maxima.  Ties (a) inside the top 200, (b) straddling the 200th place (lowest indices kept) and (c) none at all exercise
    the kernel's 32-bit rank with its collision check and the 64-bit (key, ~index) fallback.  Stage lists and scores against the oracle."""
    rng = np.random.default_rng(33)
    base = S.make_latent(rng, n_tex_lo=330, n_tex_hi=360)
    R = S.make_mate(rng, cb, base, frac=0.7, n_tex=600)
    ocb = oracle.codebook(codebook_bytes)
    m = M.Matcher(codebook_bytes, taps=True); m.gallery_add_dat(T.write_rolled(R)); m.gallery_commit(0)
    hr, _ = oracle.rolled(T.write_rolled(R))
    t0 = base.tex[0]
    n = len(t0.x)
    for ci, groups in enumerate(([], [(10, 40)], [(5, 330)], [(50, 90), (120, 300)])):
        des = t0.des.copy()
        for lo, hi in groups:
            des[lo:hi] = des[lo]
        L = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(t0.x, t0.y, t0.ori, des=des)])
        hl, _ = oracle.latent(ocb, T.write_latent(L))
        for stage in (0, 1, 2):
            want = oracle.trace(ocb, hl, hr, which=0, stage=stage, tie_mode=1)
            got = m.debug_stage_list(L, 0, 0, stage)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (ci, stage, got[1][:12], want[1][:12])
            assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (ci, stage)
        if groups:                                                     # the ties are really there
            sim0 = oracle.trace(ocb, hl, hr, which=0, stage=0, tie_mode=1)[0]
            assert len(np.unique(sim0)) < len(sim0) - 10, ci
        rc, want_sc = oracle.pair(ocb, hl, hr, 1)
        got_sc = m.search([L], k=0, want_parts=True)["parts"][0, 0]
        assert np.array_equal(got_sc.view(np.uint32), want_sc[:4].view(np.uint32)), (ci, got_sc, want_sc)
        oracle.lib.orc_latent_free(hl)
    m.close()


def test_template_shapes_sweep():
    """Correspondence lists of every length (2-120 minutiae, 20-1000 texture points per template: short lists, nearly empty last row
    blocks, lists below and above the top-120 / top-200 cuts): tools/shape_sweep.py compares every part score of 8 x 30 pairs with the
    oracle, bit for bit (the wide run behind DESIGN section 2 is the same script with more seeds)."""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shape_sweep.py"), "5", "8", "30"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "differing bit: 0" in out.stdout, (out.stdout[-600:], out.stderr[-600:])


def test_python_drivers_equal_cli(codebook_bytes, cb, small, tmp_path):
    """host/matcher.py's One2List_matching / List2List_matching (the reference's two drivers, matcher.h:44-51) write the files the
    `match` binary writes, from a directory and from a packed container."""
    import os, subprocess
    lats, gal = small
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
    for d in ("gal", "lat", "cli", "py", "py2", "work"):
        (tmp_path / d).mkdir()
    for j, g in enumerate(gal[:9]):
        (tmp_path / "gal" / f"R{j:03d}.dat").write_bytes(T.write_rolled(g))
    (tmp_path / "gal" / "R_empty.dat").write_bytes(b"")
    for i, L in enumerate(lats[:2]):
        (tmp_path / "lat" / f"L{i}.dat").write_bytes(T.write_latent(L))
    (tmp_path / "lat" / "Lnone.dat").write_bytes(T.write_latent(T.FPTemplate()))       # a latent without any template: score file `0` (matcher.cpp:153-163)
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    box = str(tmp_path / "g.afisgal")
    common = ["-g", str(tmp_path / "gal"), "-c", str(cbp), "-s", str(tmp_path / "cli") + "/"]
    for mode in (["-ldir", str(tmp_path / "lat")], ["-l", str(tmp_path / "lat" / "L1.dat"), "-pack", box]):
        o = subprocess.run([exe] + mode + common, capture_output=True, text=True, cwd=tmp_path / "work")
        assert o.returncode == 0, o.stderr
    cli = {n: (tmp_path / "cli" / n).read_text() for n in os.listdir(tmp_path / "cli")}
    # the -l run came last, so L1.csv holds the rank list; L0.csv the List2List scores
    for src, out in ((str(tmp_path / "gal"), "py"), (box, "py2")):
        m = M.Matcher(str(cbp))
        files = m.load_gallery_dir(src)
        assert len(files) == 10
        assert m.List2List_matching(str(tmp_path / "lat"), str(tmp_path / out) + "/") == 0
        assert (tmp_path / out / "Lnone.csv").read_text() == "0\n" == cli["Lnone.csv"]
        os.remove(tmp_path / out / "Lnone.csv")
        assert m.One2List_matching(str(tmp_path / "lat" / "Lnone.dat"), str(tmp_path / out) + "/") == 1        # :260-268 then :296-300
        assert (tmp_path / out / "Lnone.csv").read_text() == "0\n"
        l0 = (tmp_path / out / "L0.csv").read_text()
        assert sorted(l0.splitlines()) == sorted(cli["L0.csv"].splitlines())          # directory order may differ between the two listings
        assert m.One2List_matching(str(tmp_path / "lat" / "L1.dat"), str(tmp_path / out) + "/") == 0
        got = {n: (tmp_path / out / n).read_text() for n in os.listdir(tmp_path / out)}
        assert got["L1.csv"] == cli["L1.csv"]
        corr = [n for n in cli if n.startswith("corr")]
        assert corr and sorted(n for n in got if n.startswith("corr")) == sorted(corr)
        for n in corr:
            assert got[n] == cli[n], n
        m.close()


@pytest.mark.parametrize("rolled_minu", [None, (200, 400, 600)])
def test_edge_fusion_rules(codebook_bytes, cb, oracle, rolled_minu):
    """Template selection and fusion (matcher.cpp:376-417, :188) on latents that lack some of templates 26 / 2 / 11, texture, or everything.  rolled_minu: the same
    latents against rolled templates of 200 / 400 / 600 minutiae — shapes the small class of the candidate kernel does not take, so that a launch has EMPTY latent
    lists next to lists of the medium / large class and of the any-shape kernel (k_minu_classify must count an empty list for no class)."""
    base, variants = cases.edge_latents(cb)
    rng = np.random.default_rng(11)
    nm = rolled_minu or (None, None, None)
    kw = [dict(n_minu=n) if n else {} for n in nm]
    gal = [S.make_mate(rng, cb, base, frac=0.8, n_tex=400, **kw[0]), S.make_mate(rng, cb, base, frac=0.4, n_tex=350, **kw[1]), S.make_rolled(rng, cb, n_tex=300, **kw[2])]
    no_tex = T.FPTemplate(minu=list(gal[0].minu), tex=[])
    no_minu = T.FPTemplate(minu=[], tex=list(gal[0].tex))
    empty = T.FPTemplate()
    gal += [no_tex, no_minu, empty]
    m = M.Matcher(codebook_bytes)
    for g in gal:
        # a rolled template without minutiae cannot be written by the reference's writer (it stops after the header), add by view
        m.gallery_add([g])
    m.gallery_commit(0)
    ocb = oracle.codebook(codebook_bytes)
    names = list(variants)
    res = m.search([variants[n] for n in names], k=0, want_parts=True)
    for qi, n in enumerate(names):
        L = variants[n]
        hl = oracle.latent(ocb, T.write_latent(L))[0] if L.minu else None
        for gi, g in enumerate(gal):
            if hl is None or not g.minu:
                continue            # not expressible as a .dat for the oracle's parser; covered by the expectations below
            hr = oracle.rolled(T.write_rolled(g))[0]
            rc, want = oracle.pair(ocb, hl, hr, 1)
            if rc == 1:
                assert res["status"][qi] == 1 and res["scores"][qi, gi] == -1.0
                continue
            assert res["status"][qi] == 0
            got = np.append(res["parts"][qi, gi], res["scores"][qi, gi])
            if rc == 2:
                assert res["scores"][qi, gi] == -1.0
                continue
            # the oracle reports the texture score in slot 3 regardless of where the reference stores it
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, gi, got, want)      # bit for bit
    if rolled_minu:                                            # every task accounted for, the larger classes and the any-shape kernel all used; all-template mode on the same shapes
        tm = m.timing()
        assert tm["minu_tasks_small"] + tm["minu_tasks_medium"] + tm["minu_tasks_large"] + tm["minu_fallback_tasks"] <= tm["minu_tasks"]
        assert tm["minu_tasks_medium"] > 0 and tm["minu_tasks_large"] > 0 and tm["minu_fallback_tasks"] > 0, tm
        for n in ("minu12_tex", "minu2_tex", "full28"):
            L = variants[n]
            qs, rs, sc = m.One2One_matching_all_templates(L)
            hl = oracle.latent(ocb, T.write_latent(L))[0]
            width = len(L.minu) + len(L.tex)
            for gi in range(4):                                # the rolled templates a .dat can express
                rc, want = oracle.all_templates(ocb, hl, oracle.rolled(T.write_rolled(gal[gi]))[0], width, tie_mode=1)
                assert np.array_equal(sc[gi].view(np.uint32), want.view(np.uint32)), (n, gi, sc[gi], want)
    # structural expectations that do not need the oracle
    q28 = names.index("full28")
    assert res["scores"][q28, 5] == -1.0                       # empty rolled template (matcher.cpp:184-187)
    assert res["parts"][q28, 3, 3] == 0.0                      # rolled without texture
    assert np.all(res["parts"][q28, 4, :3] == 0.0)             # rolled without minutiae
    q0 = names.index("minu0_tex")
    assert res["scores"][q0, 0] == res["parts"][q0, 0, 3]                          # texture at score[0], weight 1 (exactly)
    q27 = names.index("minu27_tex")
    p = res["parts"][q27, 0]
    assert res["scores"][q27, 0] == np.float32(np.float32(p[0] + p[1]) + p[2])     # score[28] out of range -> 0 (exactly)


# ---- committed golden vectors --------------------------------------------------------------------------------------------
def test_golden_vectors(codebook_bytes):
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_pairs.npz"))
    m = M.Matcher(codebook_bytes)
    for j in range(12):
        assert m.gallery_add_dat(gold[f"rolled_{j}"].tobytes()) == 0
    m.gallery_commit(0)
    res = m.search_dat([gold[f"latent_{i}"].tobytes() for i in range(2)], k=3, want_parts=True)
    got = np.concatenate([res["parts"], res["scores"][..., None]], axis=-1)
    want = gold["parts"][1]                                   # tie_mode 1: equal keys by ascending index
    diff = got.view(np.uint32) != want.view(np.uint32)           # the committed vectors are reproduced bit for bit
    assert not diff.any(), (np.argwhere(diff), got[diff], want[diff])
    assert list(res["topk_idx"][0]) == [0, 1, 2] and list(res["topk_idx"][1]) == [3, 4, 5]
    # ... and with option ref_tie_order 2 the OTHER committed set: tie_mode 0, std::sort at every site — the order the reference binary executes (six of the 24 pairs differ between the sets)
    m.set_option("ref_tie_order", 2)
    res = m.search_dat([gold[f"latent_{i}"].tobytes() for i in range(2)], k=3, want_parts=True)
    got = np.concatenate([res["parts"], res["scores"][..., None]], axis=-1)
    want0 = gold["parts"][0]
    assert int((want0.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum()) >= 4
    diff = got.view(np.uint32) != want0.view(np.uint32)
    assert not diff.any(), (np.argwhere(diff), got[diff], want0[diff])
    m.close()


# ---- size-independent properties at a larger scale --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def medium(cb):
    G, Q = 3000, 6
    lats = S.make_latents(77, Q)
    gal = S.make_packed_gallery(77, G, cb)
    planted = S.plant_mates(77, gal, cb, lats)
    return lats, gal, planted


def test_medium_properties_and_sharding(codebook_bytes, cb, oracle, medium):
    lats, gal, planted = medium
    G = gal.G
    m = M.Matcher(codebook_bytes, taps=True)                            # the test library: step (5) compares with the direct reference kernels
    m.gallery_add_packed(gal); m.gallery_commit(0)
    r1 = m.search(lats, k=24, want_parts=True)
    mp = M.Matcher(codebook_bytes); mp.gallery_add_packed(gal); mp.gallery_commit(0)      # the product library gives the same bits
    rp = mp.search(lats, k=24, want_parts=True); mp.close()
    assert np.array_equal(rp["parts"].view(np.uint32), r1["parts"].view(np.uint32)) and np.array_equal(rp["topk_idx"], r1["topk_idx"])
    # (1) idempotence + resident == one-shot
    qh = m.upload_queries(lats)
    r2 = m.search_resident(qh, k=24, want_scores=True, want_parts=True)
    m.free_queries(qh)
    assert np.array_equal(r1["scores"], r2["scores"]) and np.array_equal(r1["topk_idx"], r2["topk_idx"])
    # (2) the planted mates lead every rank list in planting order (overlap 0.8 > 0.5 > 0.35 > 0.25)
    for q in range(len(lats)):
        want = [g for g, _ in planted[q]]
        assert list(r1["topk_idx"][q][:len(want)]) == want, (q, r1["topk_idx"][q][:6], want)
        assert r1["topk_score"][q][0] > 50
    # (3) the rank list is the top-k of the score vector, score descending then index ascending
    for q in range(len(lats)):
        order = np.lexsort((np.arange(G), -r1["scores"][q].astype(np.float64)))[:24]
        assert np.array_equal(r1["topk_idx"][q], order) and np.array_equal(r1["topk_score"][q], r1["scores"][q][order])
    # (4) fusion identity on every pair: final = (s0+s1)+s2 + 0.3*tex  (28 latent minutiae templates)
    p = r1["parts"]
    fused = ((p[..., 0] + p[..., 1]) + p[..., 2]).astype(np.float64) + p[..., 3].astype(np.float64) * 0.3
    assert np.array_equal(fused.astype(np.float32), r1["scores"])
    assert (r1["scores"] >= 0).all()
    # (5) every ADC variant and the generic minutiae candidate kernel give identical bits
    for v in (0, 1, 6, 8):                                              # 8 = 16-bit bound pass + exact refine
        m.set_option("adc_variant", v)
        r0 = m.search(lats, k=0)
        assert np.array_equal(r0["scores"], r1["scores"]), v
    m.set_option("adc_variant", 7); m.set_option("minu_generic", 1)
    r0 = m.search(lats, k=0)
    assert np.array_equal(r0["scores"], r1["scores"])
    m.set_option("minu_generic", 0)
    # (5b) the ADC chunk size only shapes the launch (auto: ~640 templates, a multiple of 8 chunks): odd sizes, one template per
    # workgroup and one chunk for the whole shard give the same bits, on the bound + refine kernel and on the direct one
    for v in (8, 7):
        m.set_option("adc_variant", v)
        for chunk in (1, 7, 33, 500, G):
            m.set_option("chunk", chunk)
            assert np.array_equal(m.search(lats[:3], k=0)["scores"], r1["scores"][:3]), (v, chunk)
    m.set_option("chunk", 0); m.set_option("adc_variant", 8)
    # (6) gallery sharding: two contiguous shards with global indices, merged rank lists == single-shard rank lists
    SH = importlib.import_module("msu-latentafis_amd.host.sharding")
    nm, nt = S.gallery_counts(77, G)
    bounds = SH.shard_bounds(nt, 2)
    idx, sc = [], []
    for lo, hi in bounds:
        ms = M.Matcher(codebook_bytes)
        ms.gallery_add_packed(gal.slice(lo, hi)); ms.gallery_commit(lo)
        rs = ms.search(lats, k=24)
        assert np.array_equal(rs["scores"], r1["scores"][:, lo:hi])
        idx.append(rs["topk_idx"]); sc.append(rs["topk_score"])
        ms.close()
    mi, msc = SH.merge_topk(np.stack(idx), np.stack(sc), 24)
    assert np.array_equal(mi, r1["topk_idx"]) and np.array_equal(msc, r1["topk_score"])
    # (7) a sample of pairs against the oracle (mates + random non-mates)
    ocb = oracle.codebook(codebook_bytes)
    rng = np.random.default_rng(0)
    bad = 0; n = 0
    for q in range(2):
        hl, _ = oracle.latent(ocb, T.write_latent(lats[q]))
        sample = [g for g, _ in planted[q]] + list(rng.integers(0, G, 40))
        for g in sample:
            hr, _ = oracle.rolled(T.write_rolled(gal.template(int(g))))
            rc, want = oracle.pair(ocb, hl, hr, 1)
            got = np.append(r1["parts"][q, g], r1["scores"][q, g])
            n += 1
            bad += int((got.view(np.uint32) != want.view(np.uint32)).any())
    assert bad == 0, (bad, n)


def test_cli_matches_oracle(codebook_bytes, cb, oracle, small, tmp_path):
    """The drop-in CLI: -l and -ldir output files against the oracle's scores (SURVEY §3.1 / §3.2 formats)."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(M.LIB_PATH), "match"], check=True)
    lats, gal = small
    (tmp_path / "work").mkdir(); (tmp_path / "gal").mkdir(); (tmp_path / "lat").mkdir(); (tmp_path / "out").mkdir()
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    for j, g in enumerate(gal[:12]):
        (tmp_path / "gal" / f"R{j:03d}.dat").write_bytes(T.write_rolled(g))
    (tmp_path / "gal" / "R_empty.dat").write_bytes(b"")          # empty file -> score -1
    for i, L in enumerate(lats[:2]):
        (tmp_path / "lat" / f"L{i}.dat").write_bytes(T.write_latent(L))
    ocb = oracle.codebook(codebook_bytes)
    out = subprocess.run([exe, "-ldir", str(tmp_path / "lat"), "-g", str(tmp_path / "gal"), "-s", str(tmp_path / "out") + "/", "-c", str(cbp)],
                         capture_output=True, text=True, cwd=tmp_path / "work")
    assert out.returncode == 0, out.stderr
    assert "Gallery size: 13" in out.stdout and "Latent minutiae templates: 28" in out.stdout and "Total matching duration (ms):" in out.stdout
    for i in range(2):
        hl, _ = oracle.latent(ocb, T.write_latent(lats[i]))
        lines = (tmp_path / "out" / f"L{i}.csv").read_text().splitlines()
        assert len(lines) == 13
        for line in lines:
            path, score = line.rsplit(",", 1)
            path = path.strip('"')
            hr, _ = oracle.rolled(open(path, "rb").read())
            rc, want = oracle.pair(ocb, hl, hr, 1)
            exp = -1.0 if rc == 2 else want[4]
            assert score == "%.3f" % exp, (line, exp)             # the printed digits, not a tolerance (matcher.cpp:201-204)
    out = subprocess.run([exe, "-l", str(tmp_path / "lat" / "L0.dat"), "-g", str(tmp_path / "gal"), "-s", str(tmp_path / "out") + "/", "-c", str(cbp)],
                         capture_output=True, text=True, cwd=tmp_path / "work")
    assert out.returncode == 0, out.stderr
    lines = (tmp_path / "out" / "L0.csv").read_text().splitlines()
    assert lines[0] == "filename,score" and len(lines) == 14 and lines[1].startswith('1"') and "R000.dat" in lines[1]
    assert "Match Results" in out.stdout and "Rank     Filename      Score" in out.stdout
    scores = [float(l.rsplit(",", 1)[1]) for l in lines[1:]]
    assert scores == sorted(scores, reverse=True) and scores[-1] == -1.0
    # correspondence files of the ranked templates (matcher.cpp:321-327, :497-505): <score dir>/corr<latent>_<rolled>_<i>.csv
    hl, _ = oracle.latent(ocb, T.write_latent(lats[0]))
    sel = (26, 2, 11)
    n_lines = 0
    for line in lines[1:]:
        path = line.rsplit(",", 1)[0].split('"')[1]
        stem = os.path.splitext(os.path.basename(path))[0]
        hr, _ = oracle.rolled(open(path, "rb").read())
        for i in range(3):
            f = tmp_path / "out" / f"corrL0_{stem}_{i}.csv"
            tr = oracle.trace(ocb, hl, hr, which=i + 1, stage=2, tie_mode=1) if os.path.getsize(path) > 10 else None
            if tr is None:
                assert not f.exists(), f
                continue
            _, li, ri = tr
            _, R = T.read_rolled(open(path, "rb").read())
            lm, rm = lats[0].minu[sel[i]], R.minu[0]
            want = [f"{lm.x[a]},{lm.y[a]},{rm.x[b]},{rm.y[b]}" for a, b in zip(li, ri)]
            assert f.read_text().splitlines() == want, f
            n_lines += len(want)
    assert n_lines > 0


def test_edge_shapes_against_oracle(codebook_bytes, cb, oracle):
    """Shapes off the fast paths: > 64 latent / > 128 rolled minutiae (generic candidate kernel), texture templates above the
    1000-point clamp (matcher.cpp:544-547), tiny templates (fewer than 120 / 200 candidates), duplicated points (ties)."""
    rng = np.random.default_rng(42)
    big = S.make_latent(rng, n_tex_lo=1100, n_tex_hi=1200, n_minu_lo=90, n_minu_hi=110)        # texture > 1000 rows, minutiae > 64
    tiny = S.make_latent(rng, n_tex_lo=40, n_tex_hi=60, n_minu_lo=3, n_minu_hi=6)
    dup = S.make_latent(rng, n_tex_lo=230, n_tex_hi=260)
    t0 = dup.tex[0]
    t0.x[50:100] = t0.x[0:50]; t0.y[50:100] = t0.y[0:50]; t0.des[50:100] = t0.des[0:50]; t0.ori[50:100] = t0.ori[0:50]   # exact duplicates
    lats = [big, tiny, dup]
    gal = []
    for L in lats:
        gal.append(S.make_mate(rng, cb, L, frac=0.8, n_minu=min(300, max(8, len(L._pool[0]) * 3)), n_tex=1300 if L is big else 500))
        gal.append(S.make_mate(rng, cb, L, frac=0.4, n_minu=150, n_tex=700))
    gal.append(S.make_rolled(rng, cb, n_minu=5, n_tex=30))
    gal.append(S.make_rolled(rng, cb, n_minu=260, n_tex=1900))
    m = M.Matcher(codebook_bytes)
    m.gallery_add(gal); m.gallery_commit(0)
    res = m.search(lats, k=0, want_parts=True)
    ocb = oracle.codebook(codebook_bytes)
    hr = [oracle.rolled(T.write_rolled(g))[0] for g in gal]
    worst = 0.0
    for qi, L in enumerate(lats):
        hl, _ = oracle.latent(ocb, T.write_latent(L))
        rc, sc, parts = oracle.search(ocb, hl, hr, tie_mode=1, want_parts=True)
        got = np.concatenate([res["parts"][qi], res["scores"][qi][:, None]], axis=1)
        err = np.abs(got - parts) / np.maximum(1.0, np.abs(parts))
        worst = max(worst, float(err.max()))
        diff = got.view(np.uint32) != parts.view(np.uint32)
        assert not diff.any(), (qi, np.argwhere(diff), got[diff], parts[diff])
    assert res["scores"][0, 0] > 100 and res["scores"][2, 4] > 50
    print("edge shapes: max rel err", worst)


def test_angle_stage_atan2_equals_libm_on_every_coordinate_difference(codebook_bytes, oracle):
    """The angle tests of matcher.cpp:1503-1549 compare line angles -atan2f(dy, dx) with pi/6 in double; the device evaluates
    atan2 in double and rounds to float.  Its arguments are differences of integer point coordinates, so the claim "same bits
    as the CPU's atan2f" is checked here EXHAUSTIVELY over every (dy, dx) with |dy|, |dx| <= 2047 — the whole range of
    minutiae pixel coordinates (images up to 2048 px; the generator's are 768 x 800) and of texture block coordinates."""
    R = 2047
    m = M.Matcher(codebook_bytes, taps=True)
    got = m.debug_atan2_grid(R)
    m.close()
    want = oracle.atan2f_grid(R)
    diff = got.view(np.uint32) != want.view(np.uint32)
    assert not diff.any(), (int(diff.sum()), np.argwhere(diff)[:5] - R, got[diff][:5], want[diff][:5])


def test_distance_stage_packed_arithmetic_equals_the_plain_evaluation(codebook_bytes):
    """S8 on the packed paths (csrc/graph_arith.h) takes two shortcuts around matcher.cpp:1246-1272 / :1372-1393: RN(sqrt n) of the
    integer n = dx^2 + dy^2 from v_rsq_f32 + one fma step, and "H != 0" (dist < 30) from n1, n2 without square roots outside a guard
    band.  Both are compared on the device with the plain evaluation (correctly rounded roots, float subtraction, compare): every
    integer up to 2 * 2047^2; every texture pair of [0, 4802]^2 (|d| < 50 blocks per axis, matcher.cpp:1257); 4e8 minutiae pairs
    within +-12 of the 30 px threshold.  Not one root and not one decision may differ; the band must stay a rare case."""
    m = M.Matcher(codebook_bytes, taps=True)
    c = m.debug_graph_arith()
    m.close()
    assert c[0] == 0, c
    assert c[1] == 4803 * 4803 and c[3] == 0 and c[2] < 1e-3 * c[1], c
    assert c[4] > 4e8 and c[6] == 0, c


def test_rank_lists_device_kernel_equals_host_sort(codebook_bytes, cb, medium, small):
    """S11 (matcher.cpp:306-309 + the documented tie rule): k <= 64 is ranked by the device kernel, k > 64 by a host partial_sort of the
    copied score matrix; both must give the lexsort (score descending, index ascending) — including the 99 % of pairs tied at 0 —
    and the -1 / -inf padding when k exceeds the gallery."""
    lats, gal, planted = medium
    G = gal.G
    m = M.Matcher(codebook_bytes)
    m.gallery_add_packed(gal); m.gallery_commit(1000)                      # index_base: rank lists carry global indices
    qh = m.upload_queries(lats)
    r64 = m.search_resident(qh, k=64, want_scores=True)
    r100 = m.search_resident(qh, k=100)                                    # host path
    r1 = m.search_resident(qh, k=1)
    m.free_queries(qh)
    assert (r64["scores"] == 0).mean() > 0.3                                # the tie at zero is massive
    for q in range(len(lats)):
        order = np.lexsort((np.arange(G), -r64["scores"][q].astype(np.float64)))
        assert np.array_equal(r64["topk_idx"][q], order[:64] + 1000) and np.array_equal(r64["topk_score"][q], r64["scores"][q][order[:64]])
        assert np.array_equal(r100["topk_idx"][q], order[:100] + 1000)
        assert r1["topk_idx"][q, 0] == order[0] + 1000
    m.close()
    lats2, gal2 = small
    m = _matcher(codebook_bytes, gal2[:12])
    r = m.search(lats2, k=24)
    assert (r["topk_idx"][:, 12:] == -1).all() and np.isneginf(r["topk_score"][:, 12:]).all() and (r["topk_idx"][:, :12] >= 0).all()
    for q in range(len(lats2)):
        assert sorted(r["topk_idx"][q, :12]) == list(range(12))
    m.close()


def test_the_tolerance_path_is_gone(codebook_bytes, cb, small):
    """Rounds 1-2 shipped an opt-in 16-bit LUT path that was NOT bit-exact and missed SURVEY 8d's tolerance (98.2 % of pairs within 1e-3 instead of
    99.9 %).  It was removed: lut_dtype accepts only 32, every ADC variant left is bit-identical to the reference arithmetic."""
    lats, gal = small
    m = _matcher(codebook_bytes, gal)
    m.set_option("lut_dtype", 32)
    with pytest.raises(M.AfisError):
        m.set_option("lut_dtype", 16)
    m.close()


def _orc_pair(oracle, ocb, lat, rolled):
    """Oracle handles of one (latent, rolled) pair, from the .dat bytes the reference's own reader would parse."""
    return oracle.latent(ocb, T.write_latent(lat))[0], oracle.rolled(T.write_rolled(rolled))[0]


def _same_bits(a, b):
    """Bit equality of two float arrays, NaN counted as equal to NaN (the payload of a NaN is not part of the contract)."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def test_bound_and_refine_kernel_equals_direct_kernel_row_by_row(codebook_bytes, cb, oracle, medium):
    """adc_variant 8 (16-bit bound pass + exact evaluation of the candidates) and 9 (fp16 matrix-core bound pass + recomputation) against the ORACLE
    (matcher.cpp:563-595, :723-735 restated) and the direct exact kernel (7): every row maximum and every first arg-max of 6 latents x 40 gallery
    templates (about 160 000 rows), bit for bit."""
    lats, gal, planted = medium
    m = M.Matcher(codebook_bytes, taps=True)
    m.gallery_add_packed(gal); m.gallery_commit(0)
    ocb = oracle.codebook(codebook_bytes)
    rng = np.random.default_rng(8)
    n = 0
    for qi in range(len(lats)):
        gs = [g for g, _ in planted[qi]] + [int(x) for x in rng.integers(0, gal.G, 36)]
        for g in gs:
            m.set_option("adc_variant", 7); v7, a7 = m.debug_texture_rowmax(lats[qi], g)
            hl, hr = _orc_pair(oracle, ocb, lats[qi], gal.template(g))
            ov, oa = oracle.texture_rowmax(ocb, hl, hr)
            assert np.array_equal(v7.view(np.uint32), ov.view(np.uint32)) and np.array_equal(a7, oa), (7, qi, g)
            for v in (8, 9):                                            # 8: 16-bit LDS-table bound pass; 9: fp16 matrix-core bound pass
                m.set_option("adc_variant", v); v8, a8 = m.debug_texture_rowmax(lats[qi], g)
                assert np.array_equal(v7.view(np.uint32), v8.view(np.uint32)) and np.array_equal(a7, a8), (v, qi, g, np.argwhere(a7 != a8)[:4])
            n += len(v7)
    m.close()
    assert n > 100000


def test_bound_and_refine_kernel_on_ties_and_near_ties(codebook_bytes, cb, oracle):
    """The bound logic at its edges: rolled templates whose points are all identical (every quantised sum ties: each lane's second
    sum equals its best, so the whole row is evaluated exactly and the FIRST point must win), alternate between two codes, or differ
    from one another in a single sub-quantizer (exact similarities closer than the quantisation step); latent rows with zero, tiny and
    duplicated descriptors (degenerate table ranges).  Variants 8 and 9 must reproduce the ORACLE (the reference's table arithmetic and std::max_element,
    matcher.cpp:563-595, :723-735) bit for bit, values and first arg-maxima; the direct exact kernel (7) is the second witness."""
    rng = np.random.default_rng(21)
    base = S.make_latent(rng, n_tex_lo=330, n_tex_hi=360)
    lt = base.tex[0]
    des = lt.des.copy()
    des[0] = 0.0; des[1] = 1e-6 * des[1]; des[2] = des[3]; des[5:9] = des[4]
    lat = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(lt.x, lt.y, lt.ori, des=des)])
    def rolled(codes):
        r = S.make_rolled(rng, cb, n_tex=len(codes))
        r.tex[0].codes[:] = codes
        return r
    n = 700
    same = np.tile(rng.integers(0, 256, (1, 16)).astype(np.uint8), (n, 1))
    two = np.where((np.arange(n) % 2 == 0)[:, None], same, rng.integers(0, 256, (1, 16)).astype(np.uint8))
    near = np.tile(cb.encode(des[10:11]), (n, 1)).astype(np.uint8)
    near[np.arange(n), rng.integers(0, 16, n)] = rng.integers(0, 256, n).astype(np.uint8)           # one sub-quantizer off the row's own best code
    near2 = near.copy(); near2[::3] = near[0]                                                     # many exact duplicates among near ties
    enc = np.repeat(cb.encode(des[:350]), 2, axis=0)[:n].astype(np.uint8)                         # every latent row's best code, twice
    gal = [rolled(same), rolled(two), rolled(near), rolled(near2), rolled(enc), rolled(rng.integers(0, 256, (65, 16)).astype(np.uint8)),
           rolled(rng.integers(0, 256, (1, 16)).astype(np.uint8))]
    m = _matcher(codebook_bytes, gal, taps=True)
    ocb = oracle.codebook(codebook_bytes)
    want_parts = []
    for g in range(len(gal)):
        hl, hr = _orc_pair(oracle, ocb, lat, gal[g])
        ov, oa = oracle.texture_rowmax(ocb, hl, hr)
        want_parts.append(oracle.pair(ocb, hl, hr, 1)[1][:4])
        for v in (7, 8, 9):
            m.set_option("adc_variant", v); v8, a8 = m.debug_texture_rowmax(lat, g)
            assert np.array_equal(ov.view(np.uint32), v8.view(np.uint32)), (v, g, np.argwhere(ov != v8)[:4])
            assert np.array_equal(oa, a8), (v, g, np.argwhere(oa != a8)[:4], oa[:8], a8[:8])
    assert (m.debug_texture_rowmax(lat, 0)[1] == 0).all()                                          # all points identical: the first one
    for v in (7, 8, 9):
        m.set_option("adc_variant", v); r8 = m.search([lat], k=0, want_parts=True)
        assert np.array_equal(np.asarray(want_parts, np.float32).view(np.uint32), r8["parts"][0].view(np.uint32)), v     # per-part scores of the oracle
    m.close()


def test_bound_and_refine_kernel_with_unnormalised_latent_descriptors(codebook_bytes, cb, oracle):
    """The candidate margin of the bound pass comes from each row's own table (k_lutq_build), not from an assumed descriptor norm of 1.73:
    latent texture descriptors scaled by 8 and 40, shifted by +3, and mixed per row (table entries up to ~10^5: fp32 rounding errors far
    above the normalised case's 2e-6) against random and near-tie rolled templates.  Variants 8 and 9 must equal the ORACLE bit for bit, values
    and first arg-maxima (the direct exact kernel, 7, is the second witness)."""
    rng = np.random.default_rng(77)
    base = S.make_latent(rng, n_tex_lo=300, n_tex_hi=340)
    lt = base.tex[0]
    def rolled(codes):
        r = S.make_rolled(rng, cb, n_tex=len(codes))
        r.tex[0].codes[:] = codes
        return r
    n = 900
    near = np.tile(cb.encode(lt.des[7:8]), (n, 1)).astype(np.uint8)
    near[np.arange(n), rng.integers(0, 16, n)] = rng.integers(0, 256, n).astype(np.uint8)
    gal = [rolled(rng.integers(0, 256, (n, 16)).astype(np.uint8)), rolled(near), rolled(rng.integers(0, 256, (130, 16)).astype(np.uint8))]
    m = _matcher(codebook_bytes, gal, taps=True)
    ocb = oracle.codebook(codebook_bytes)
    per_row = np.where(np.arange(lt.n)[:, None] % 3 == 0, 1.0, np.where(np.arange(lt.n)[:, None] % 3 == 1, 17.0, 0.01)).astype(np.float32)
    huge = lt.des.copy(); huge[::5] *= np.float32(3000)                 # beyond what fp16 operands carry: variant 9 evaluates those rows over every point
    for name, des in (("x8", lt.des * np.float32(8)), ("x40", lt.des * np.float32(40)), ("+3", lt.des + np.float32(3)), ("mixed", lt.des * per_row), ("huge", huge)):
        lat = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(lt.x, lt.y, lt.ori, des=np.ascontiguousarray(des, np.float32))])
        want_parts = []
        for g in range(len(gal)):
            hl, hr = _orc_pair(oracle, ocb, lat, gal[g])
            ov, oa = oracle.texture_rowmax(ocb, hl, hr)
            want_parts.append(oracle.pair(ocb, hl, hr, 1)[1][:4])
            for v in (7, 8, 9):
                m.set_option("adc_variant", v); v8, a8 = m.debug_texture_rowmax(lat, g)
                assert np.array_equal(ov.view(np.uint32), v8.view(np.uint32)), (v, name, g, np.argwhere(ov != v8)[:4], ov[:3], v8[:3])
                assert np.array_equal(oa, a8), (v, name, g, np.argwhere(oa != a8)[:4])
        for v in (7, 9):
            m.set_option("adc_variant", v); r9 = m.search([lat], k=0, want_parts=True)
            assert np.array_equal(np.asarray(want_parts, np.float32).view(np.uint32), r9["parts"][0].view(np.uint32)), (v, name)
    m.close()


def test_bound_pass_with_nan_and_inf_latent_descriptors(codebook_bytes, cb, oracle):
    """Latent texture rows holding NaN, +inf, -inf, fp32 values beyond fp16's range (7e4, 3e38) and denormals.  fp16 operands cannot carry such rows:
    k_mf_rows forces them (Tg = Es = inf) and k_tex_refine evaluates them over every point in the reference's fp32 arithmetic, where include.h:327-359 /
    matcher.cpp:571-592, :730 define the result: a NaN table entry makes every similarity of the row NaN and std::max_element keeps the FIRST point;
    an infinite one makes them all -inf, again the first point.  Row maxima / first arg-maxima of variants 7, 8 and 9 equal the oracle's (NaN == NaN),
    the finite rows around them are untouched, and a search over such a latent neither hangs nor crashes and gives the same scores in variants 7 and 9
    (the reference's own S7 sort of NaN keys is undefined behaviour: there is no oracle value for the pair score)."""
    rng = np.random.default_rng(404)
    base = S.make_latent(rng, n_tex_lo=300, n_tex_hi=320)
    lt = base.tex[0]
    des = lt.des.copy()
    des[3, 5] = np.nan; des[4, :] = np.nan; des[9, 0] = np.inf; des[10, 95] = -np.inf; des[11, 40] = np.inf; des[11, 41] = -np.inf
    des[20, 7] = 7e4; des[21, 8] = -3e38; des[22, :] = 1e-42; des[23, 17] = 1001.0; des[24, 17] = 999.0; des[25, :] = 0.0
    lat = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(lt.x, lt.y, lt.ori, des=des)])
    gal = [S.make_rolled(rng, cb, n_tex=n) for n in (640, 33, 1)] + [S.make_mate(rng, cb, base, frac=0.6, n_tex=500)]
    m = _matcher(codebook_bytes, gal, taps=True)
    ocb = oracle.codebook(codebook_bytes)
    for g in range(len(gal)):
        hl, hr = _orc_pair(oracle, ocb, lat, gal[g])
        ov, oa = oracle.texture_rowmax(ocb, hl, hr)
        assert np.isnan(ov[[3, 4]]).all() and (oa[[3, 4]] == 0).all() and np.isneginf(ov[[9, 10]]).all() and (oa[[9, 10, 11]] == 0).all()
        for v in (9, 7, 8):
            m.set_option("adc_variant", v); vv, aa = m.debug_texture_rowmax(lat, g)
            # the default path (9) reproduces the oracle's NaN rows; the alternative kernels (7, 8) start their running maximum at -inf, so a row whose
            # similarities are all NaN may read -inf there (first point, as everywhere): a documented difference on rows the reference cannot score (its S7
            # sort of NaN keys is undefined behaviour)
            if v != 9:                                                  # NaN rows of the alternative kernels: NaN (variant 8's single-candidate path) or -inf
                nan_rows = np.isnan(ov)
                assert (np.isnan(vv[nan_rows]) | np.isneginf(vv[nan_rows])).all(), (v, g, vv[nan_rows])
                vv = np.where(nan_rows, ov, vv)
                n_pts = gal[g].tex[0].n
                assert ((aa[nan_rows] >= 0) & (aa[nan_rows] < n_pts)).all(), (v, g, aa[nan_rows])     # a valid point (variant 8: not necessarily the first) — never an index S7 cannot follow
                aa = np.where(nan_rows, oa, aa)
            assert _same_bits(ov, vv), (v, g, np.argwhere(ov.view(np.uint32) != vv.view(np.uint32))[:6].ravel(), ov[[3, 9, 20, 21]], vv[[3, 9, 20, 21]])
            assert np.array_equal(oa, aa), (v, g, np.argwhere(oa != aa)[:6].ravel())
    # a search over such a latent: no hang, no fault, the minutiae parts untouched by the texture rows, the same answer twice
    m.set_option("adc_variant", 9); m.set_option("mf_stats", 1)
    r9 = m.search([lat], k=0, want_parts=True); r9b = m.search([lat], k=0, want_parts=True)
    assert _same_bits(r9["parts"], r9b["parts"]) and m.refine_stats()["bound_violations"] == 0
    clean = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(lt.x, lt.y, lt.ori, des=lt.des)])
    rc = m.search([clean], k=0, want_parts=True)
    assert np.array_equal(rc["parts"][..., :3].view(np.uint32), r9["parts"][..., :3].view(np.uint32))
    m.set_option("adc_variant", 7); r7 = m.search([lat], k=0, want_parts=True)
    assert np.array_equal(rc["parts"][..., :3].view(np.uint32), r7["parts"][..., :3].view(np.uint32))
    m.close()


def test_bound_pass_magnitude_sweep_up_to_the_forcing_threshold(codebook_bytes, cb, oracle):
    """The bound's error term Eg grows with the descriptor's magnitude; rows are forced (evaluated everywhere) only beyond |a| > 1000.  Between the two —
    descriptors scaled by 100, 300, 900 and to a largest component of exactly 999 and 1000 (fp16-representable, the largest Eg that is still trusted), and
    just beyond (1000.5: forced) — the selection must still find every row maximum: oracle row maxima / first arg-maxima bit for bit in variant 9, the
    kernel's self-check (every exact maximum inside its bounds) silent, and the per-part scores of the oracle."""
    rng = np.random.default_rng(505)
    base = S.make_latent(rng, n_tex_lo=260, n_tex_hi=280)
    lt = base.tex[0]
    n = 800
    near = np.tile(cb.encode(lt.des[5:6]), (n, 1)).astype(np.uint8)
    near[np.arange(n), rng.integers(0, 16, n)] = rng.integers(0, 256, n).astype(np.uint8)
    def rolled(codes):
        r = S.make_rolled(rng, cb, n_tex=len(codes))
        r.tex[0].codes[:] = codes
        return r
    gal = [rolled(rng.integers(0, 256, (n, 16)).astype(np.uint8)), rolled(near), S.make_mate(rng, cb, base, frac=0.7, n_tex=600)]
    m = _matcher(codebook_bytes, gal, taps=True)
    m.set_option("adc_variant", 9); m.set_option("mf_stats", 1)
    ocb = oracle.codebook(codebook_bytes)
    amax = np.abs(lt.des).max(axis=1, keepdims=True)
    cases_ = [("x100", lt.des * np.float32(100)), ("x300", lt.des * np.float32(300)), ("x900", lt.des * np.float32(900)),
              ("max999", lt.des / amax * np.float32(999)), ("max1000", lt.des / amax * np.float32(1000)), ("max1000.5", lt.des / amax * np.float32(1000.5))]
    for name, des in cases_:
        lat = T.FPTemplate(minu=list(base.minu), tex=[T.TextureTemplate(lt.x, lt.y, lt.ori, des=np.ascontiguousarray(des, np.float32))])
        m.refine_stats()                                                # reset
        want_parts = []
        for g in range(len(gal)):
            hl, hr = _orc_pair(oracle, ocb, lat, gal[g])
            ov, oa = oracle.texture_rowmax(ocb, hl, hr)
            want_parts.append(oracle.pair(ocb, hl, hr, 1)[1][:4])
            vv, aa = m.debug_texture_rowmax(lat, g)
            assert np.array_equal(ov.view(np.uint32), vv.view(np.uint32)), (name, g, np.argwhere(ov != vv)[:4].ravel())
            assert np.array_equal(oa, aa), (name, g, np.argwhere(oa != aa)[:4].ravel())
        r9 = m.search([lat], k=0, want_parts=True)
        assert np.array_equal(np.asarray(want_parts, np.float32).view(np.uint32), r9["parts"][0].view(np.uint32)), name
        st = m.refine_stats()
        assert st["bound_violations"] == 0 and st["pairs"] >= len(gal), (name, st)
        if name == "max1000.5":                                         # every row has a component beyond 1000: all rows are forced, i.e. evaluated over every point
            assert st["rows_evaluated_in_full"] == st["rows_evaluated"] == st["rows"], (name, st)
        else:
            assert st["rows_evaluated_in_full"] < 0.5 * st["rows_evaluated"], (name, st)
    m.close()


def test_results_do_not_depend_on_the_schedule(codebook_bytes, cb, medium):
    """The default schedule runs the bound pass on a stream confined to 128 CUs with the minutiae stage beside it on the other 128 (option bound_cus; launches of at least 2^16
    pairs: 24 latents x 3000 templates here).  bound_cus 0 (one stream, kernels back to back), 64, 128 (default) and 192 give the same bits in scores, per-part scores and rank lists;
    the option reads back; values off the 32-CU grid are refused; back to back afis_timing's total is the sum of its stages."""
    lats, gal, planted = medium
    m = M.Matcher(codebook_bytes)
    m.gallery_add_packed(gal); m.gallery_commit(0)
    many = list(lats) * 4
    assert len(many) * gal.G >= 65536
    assert m.get_option("bound_cus") == 128 and m.get_option("adc_variant") == 9
    want = None
    for bc in (0, 64, 128, 192):
        m.set_option("bound_cus", bc)
        assert m.get_option("bound_cus") == bc
        got = m.search(many, k=24, want_parts=True)
        tm = m.timing()
        assert tm["total_ms"] > 0 and tm["adc_bound_ms"] > 0 and tm["cands_ms"] > 0 and tm["minu_graph_ms"] > 0
        if bc == 0:
            assert abs(tm["total_ms"] - (tm["lut_ms"] + tm["adc_ms"] + tm["tex_tail_ms"] + tm["minu_ms"] + tm["fuse_ms"] + tm["topk_ms"])) < 0.05 * tm["total_ms"]
        if want is None: want = got
        for key in ("scores", "parts", "topk_idx", "topk_score"):
            assert np.array_equal(np.asarray(got[key]).view(np.uint8), np.asarray(want[key]).view(np.uint8)), (bc, key)
    for bad in (8, 100, 256, -32):
        with pytest.raises(M.AfisError):
            m.set_option("bound_cus", bad)
    assert m.get_option("bound_cus") == 192
    m.close()


def test_results_do_not_depend_on_the_launch_groups(codebook_bytes, cb, oracle, medium):
    """A search's latents are cut into launch groups (option query_batch; 0 = about five million pairs per launch, and with adc_variant 9 the cuts are placed where the bound pass's
    row groups of 768 latent texture rows are fewest: afis_queries_upload).  One latent per launch, 2, 5, 7 (ragged last group), 24 and the automatic rule give the same bits in
    scores, per-part scores and rank lists, in the overlapped schedule and back to back; one latent's row of the result equals the oracle's."""
    lats, gal, planted = medium
    m = M.Matcher(codebook_bytes)
    m.gallery_add_packed(gal); m.gallery_commit(0)
    many = list(lats) * 4
    assert len(many) * gal.G >= 65536
    want = None
    for bc in (128, 0):
        m.set_option("bound_cus", bc)
        for qb in (0, 1, 2, 5, 7, 24, 256):
            m.set_option("query_batch", qb)
            assert m.get_option("query_batch") == qb
            got = m.search(many, k=24, want_parts=True)
            if want is None: want = got
            for key in ("scores", "parts", "topk_idx", "topk_score"):
                assert np.array_equal(np.asarray(got[key]).view(np.uint8), np.asarray(want[key]).view(np.uint8)), (bc, qb, key)
    with pytest.raises(M.AfisError):
        m.set_option("query_batch", 257)
    m.set_option("query_batch", 0)
    m.close()
    qi = 1                                                               # one latent's row against the oracle (every 37th template; the planted mates of the latent with them)
    picks = sorted(set(range(0, gal.G, 37)) | {g for g, _ in planted[qi]})
    ocb = oracle.codebook(codebook_bytes)
    hr = [oracle.rolled(T.write_rolled(gal.template(t)))[0] for t in picks]
    hl, _ = oracle.latent(ocb, T.write_latent(lats[qi]))
    rc, sc, parts = oracle.search(ocb, hl, hr, tie_mode=1, want_parts=True)
    got = np.concatenate([np.asarray(want["parts"])[qi][picks], np.asarray(want["scores"])[qi][picks][:, None]], axis=1).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), parts.astype(np.float32).view(np.uint32))
    assert float(np.asarray(want["scores"])[qi][planted[qi][0][0]]) > 50


def test_bound_pass_kernel_forms_are_bit_identical(codebook_bytes, cb, oracle, small):
    """The matrix-core bound pass exists in two forms — two row blocks per wave (default) and three (mf_blocks 3: a third less LDS operand traffic per MFMA; test
    library only) — which differ only in when a wave does what (round 4's third form, a two-stage software pipeline, measured 6 % slower and was deleted in round 5).
    Row maxima / first arg-maxima against the oracle, and the searches' per-part scores and rank lists against the default form's, bit for bit; workgroup chunks of 1
    and 3 templates exercise the stage / template edges of the 4-tile stages the alternative form uses."""
    lats, gal = small
    m = _matcher(codebook_bytes, gal, taps=True)
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)
    want = m.search(lats, k=10, want_parts=True)
    for mb in (3,):
        m.set_option("mf_blocks", mb)
        for chunk in (0, 1, 3):
            m.set_option("chunk", chunk)
            got = m.search(lats, k=10, want_parts=True)
            assert np.array_equal(got["parts"].view(np.uint32), want["parts"].view(np.uint32)) and np.array_equal(got["topk_idx"], want["topk_idx"]), (mb, chunk)
        for qi in range(len(lats)):
            for g in (0, 4, 17, len(gal) - 1):
                val, arg = m.debug_texture_rowmax(lats[qi], g)
                oval, oarg = oracle.texture_rowmax(ocb, hl[qi], hr[g])
                assert np.array_equal(val.view(np.uint32), oval.view(np.uint32)) and np.array_equal(arg, oarg), (mb, qi, g)
    for bad in (4, 102):
        with pytest.raises(M.AfisError):
            m.set_option("mf_blocks", bad)
    m.close()


def test_matrix_core_bound_pass_selection_statistics(codebook_bytes, cb, medium):
    """adc_variant 9 as the search runs it (rows that cannot reach a pair's top 200 are NOT evaluated): scores equal the direct exact kernel's
    bit for bit on 6 latents x 3000 templates, and the kernel's own counters say what it did — every exact row maximum inside the bounds the
    selection used, about a third of the rows evaluated, about one candidate cell per evaluated row, next to no row evaluated in full."""
    lats, gal, planted = medium
    m = M.Matcher(codebook_bytes, taps=True)
    m.gallery_add_packed(gal); m.gallery_commit(0)
    m.set_option("adc_variant", 7); r7 = m.search(lats, k=24, want_parts=True)
    m.set_option("adc_variant", 9); m.set_option("mf_stats", 1)
    r9 = m.search(lats, k=24, want_parts=True)
    st = m.refine_stats()
    assert np.array_equal(r7["parts"].view(np.uint32), r9["parts"].view(np.uint32)) and np.array_equal(r7["topk_idx"], r9["topk_idx"])
    assert st["pairs"] == len(lats) * gal.G and st["bound_violations"] == 0, st
    assert 200 * st["pairs"] <= st["rows_evaluated"] <= 0.6 * st["rows"], st
    assert st["rows_evaluated"] <= st["cells_evaluated"] + 64 * st["rows_evaluated_in_full"] and st["cells_evaluated"] <= 1.3 * st["rows_evaluated"], st
    assert st["rows_evaluated_in_full"] <= 0.01 * st["rows_evaluated"], st
    # the parity tap evaluates every row: the bounds hold there too
    for g in [planted[0][0][0], 5, 77]:
        m.debug_texture_rowmax(lats[0], g)
    st2 = m.refine_stats()
    assert st2["bound_violations"] == 0 and st2["rows_evaluated"] == st2["rows"], st2
    m.close()


def test_matrix_core_bound_pass_tile_stage_and_chunk_edges(codebook_bytes, cb, oracle):
    """k_adc_mfma's bookkeeping at its edges: rolled texture templates of 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000 and 1900 (clamped) points and none
    at all (tiles, pairs of tiles, padding tiles, the last-pair flag), latents of 1, 31, 33, 767, 769 and 1000 texture rows plus one without texture (row blocks,
    row groups of 768, launch-group cuts), and workgroup chunks of 1, 2, 3, 7 templates (stages that end inside a pair, chunks of empty templates).
    Row maxima / arg-maxima through the parity tap equal the ORACLE's (and the direct exact kernel's), per-part scores through the search the oracle's
    texture scores and the direct kernel's parts, bit for bit."""
    rng = np.random.default_rng(909)
    sizes = [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 1900, 0, 700, 0, 5]
    gal = []
    for n in sizes:
        r = S.make_rolled(rng, cb, n_tex=max(n, 1))
        if n == 0:
            r.tex = []
        gal.append(r)
    base = S.make_latent(rng, n_tex_lo=1000, n_tex_hi=1000)
    lt = base.tex[0]
    def latent(n):
        tex = [] if n == 0 else [T.TextureTemplate(lt.x[:n].copy(), lt.y[:n].copy(), lt.ori[:n].copy(), des=lt.des[:n].copy())]
        return T.FPTemplate(minu=list(base.minu), tex=tex)
    lats = [latent(n) for n in (1, 31, 33, 767, 0, 769, 1000)]
    gal[13] = S.make_mate(rng, cb, base, frac=0.6, n_tex=700)       # one real mate so that the texture scorer has something to find
    m = _matcher(codebook_bytes, gal, taps=True)
    m.set_option("adc_variant", 7)
    want = m.search(lats, k=0, want_parts=True)
    taps = {(qi, g): m.debug_texture_rowmax(lats[qi], g) for qi in (0, 2, 3, 5, 6) for g in (0, 3, 6, 9, 10, 11, 13, 15)}
    ocb = oracle.codebook(codebook_bytes)
    for (qi, g), (v7, a7) in taps.items():                            # the witness itself is checked against the oracle on exactly these shapes
        hl, hr = _orc_pair(oracle, ocb, lats[qi], gal[g])
        ov, oa = oracle.texture_rowmax(ocb, hl, hr)
        assert np.array_equal(v7.view(np.uint32), ov.view(np.uint32)) and np.array_equal(a7, oa), (qi, g)
        osc = oracle.pair(ocb, hl, hr, 1)[1]
        assert np.array_equal(want["parts"][qi, g].view(np.uint32), osc[:4].view(np.uint32)), (qi, g, want["parts"][qi, g], osc)
    m.set_option("adc_variant", 9)
    for chunk in (0, 1, 2, 3, 7):
        m.set_option("chunk", chunk)
        got = m.search(lats, k=0, want_parts=True)
        assert np.array_equal(got["parts"].view(np.uint32), want["parts"].view(np.uint32)), (chunk, np.argwhere(got["parts"] != want["parts"])[:5])
        for (qi, g), (v7, a7) in taps.items():
            v9, a9 = m.debug_texture_rowmax(lats[qi], g)
            assert np.array_equal(v7.view(np.uint32), v9.view(np.uint32)) and np.array_equal(a7, a9), (chunk, qi, g)
    assert want["parts"][6, 13, 3] > 20
    m.close()


def test_minutiae_coordinates_beyond_the_packed_path(codebook_bytes, cb, oracle):
    """S8a arithmetic paths: pixel coordinates within [0, 2047] take the packed 16-bit predicate (v_pk_sub_i16 + v_dot2), anything larger
    — here offsets of 2040 (straddling the limit), 5000 and 30000 — the generic float arithmetic, where dx*dx + dy*dy is no longer exact.
    Stage lists of the three minutiae scorers and the scores against the oracle, bit for bit."""
    rng = np.random.default_rng(33)
    base = S.make_latent(rng, n_tex_lo=210, n_tex_hi=240)
    ocb = oracle.codebook(codebook_bytes)
    R0 = S.make_mate(rng, cb, base, frac=0.8, n_tex=300)
    for ci, (off_l, off_r, scale) in enumerate(((0, 0, 1), (2040, 1500, 1), (5000, 4000, 3), (30000, 250, 1))):
        def shift(m, off):
            x = (m.x.astype(np.int64) * scale + off).astype(np.uint16).view(np.int16)
            y = (m.y.astype(np.int64) * scale + off).astype(np.uint16).view(np.int16)
            return T.MinutiaeTemplate(x, y, m.ori, m.des)
        L = T.FPTemplate(minu=[shift(m_, off_l) for m_ in base.minu], tex=list(base.tex))
        R = T.FPTemplate(minu=[shift(R0.minu[0], off_r)], tex=list(R0.tex))
        m = M.Matcher(codebook_bytes, taps=True); m.gallery_add_dat(T.write_rolled(R)); m.gallery_commit(0)
        hl, _ = oracle.latent(ocb, T.write_latent(L)); hr, _ = oracle.rolled(T.write_rolled(R))
        for which in (1, 2, 3):
            for stage in (1, 2):
                want = oracle.trace(ocb, hl, hr, which=which, stage=stage, tie_mode=1)
                got = m.debug_stage_list(L, 0, which, stage)
                assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (ci, which, stage)
                assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (ci, which, stage)
        rc, want_sc = oracle.pair(ocb, hl, hr, 1)
        got_sc = m.search([L], k=0, want_parts=True)["parts"][0, 0]
        assert np.array_equal(got_sc.view(np.uint32), want_sc[:4].view(np.uint32)), (ci, got_sc, want_sc)
        if ci == 0:
            assert want_sc[0] > 10
        m.close()


def test_a_search_allocates_before_it_queues_and_never_again(codebook_bytes, tmp_path):
    """The allocation rule: the buffers of a search's launch groups are brought to the size of the search's largest group — what it really takes: its own latent
    texture rows (round 6; rounds 4-5 sized them for 1000 rows per latent) — BEFORE anything of the search is queued: a hipMalloc behind queued work was seen to take 0.5-0.8 s
    (profiles/r04_alloc_trace.txt), on the idle device 0.3 ms.  Observable through AFIS_ALLOC_TRACE=1 (every (re)allocation of 64 MB or more, and the point where a search starts
    queuing): no search allocates after that point, in either schedule; a search of longer latents grows the row buffers (before it queues), the same search again allocates
    nothing, fewer latents never allocate, more latents per search do."""
    import subprocess, sys, textwrap
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import importlib, sys
        sys.path.insert(0, {root!r})
        import numpy as np
        T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); M = importlib.import_module("msu-latentafis_amd.host.matcher")
        cbb = open({str(cbp)!r}, "rb").read(); cb = T.Codebook.from_bytes(cbb)
        rng = np.random.default_rng(5)
        short = [S.make_latent(rng, n_tex_lo=400, n_tex_hi=420) for _ in range(8)]
        long_ = [S.make_latent(rng, n_tex_lo=990, n_tex_hi=1000) for _ in range(8)]
        gal = S.make_packed_gallery(5, 9000, cb)
        m = M.Matcher(cbb); m.gallery_add_packed(gal); m.gallery_commit(0)
        def mark(s): sys.stderr.write("mark: " + s + "\\n"); sys.stderr.flush()
        mark("short"); a = m.search(short, k=4)
        mark("long"); b = m.search(long_, k=4)
        mark("long again"); b2 = m.search(long_, k=4)
        assert np.array_equal(b["scores"], b2["scores"])
        mark("fewer"); m.search(long_[:3], k=4)
        m.set_option("bound_cus", 0)
        mark("back to back"); c = m.search(long_, k=4)
        assert np.array_equal(b["scores"], c["scores"])
        mark("more"); m.search(short + long_, k=4)
        mark("end"); m.close()
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, AFIS_ALLOC_TRACE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    allocs, late, where, queued = {}, {}, None, False
    for line in r.stderr.splitlines():
        if line.startswith("mark: "): where = line[6:]; allocs[where] = []; late[where] = []; queued = False
        elif line.startswith("queue: "): queued = True
        elif line.startswith("alloc: ") and where is not None: (late if queued else allocs)[where].append(line)
    assert all(v == [] for v in late.values()), late               # nothing is (re)allocated once a search has started queuing
    assert len(allocs["short"]) >= 4, allocs                       # the first search of a context allocates (row maxima x 2, records, candidate lists ...)
    assert 1 <= len(allocs["long"]) <= 4, allocs                   # longer latents: the row-indexed buffers grow (row maxima x 2, records), the candidate lists do not
    assert allocs["long again"] == [] and allocs["fewer"] == [] and allocs["back to back"] == [], allocs
    assert len(allocs["more"]) >= 4, allocs                        # 16 latents per search: larger buffers, once


def test_off_envelope_shapes_against_oracle(codebook_bytes, cb, oracle):
    """A 200-pair slice of tools/offenv_sweep.py: rolled minutiae templates of 129 .. 2000 minutiae, latent ones of 65 .. 200 (matcher.cpp:788-790 allows 2000 per
    template; extraction_rolled.py:105-108 caps nothing for rolled prints), texture templates beyond the 1000-row clamp (matcher.cpp:544-547), pixel coordinates on
    both sides of 2047.  Every per-part and fused score against the oracle, bit for bit; the candidate stage must have used all of its routes (the three shape
    classes of the matrix-core kernel and the any-shape kernel) and every task must be accounted for."""
    lats, rolled, mates = S.make_offenvelope_set(5, 8, 25, cb)
    m = M.Matcher(codebook_bytes)
    for R in rolled: m.gallery_add_dat(T.write_rolled(R))
    m.gallery_commit(0)
    res = m.search(lats, k=0, want_parts=True)
    tm = m.timing()
    m.close()
    assert tm["minu_tasks"] == len(lats) * 3 * len(rolled)
    assert tm["minu_tasks_medium"] > 0 and tm["minu_tasks_large"] > 0 and tm["minu_fallback_tasks"] > 0, tm
    assert tm["minu_tasks_small"] + tm["minu_tasks_medium"] + tm["minu_tasks_large"] + tm["minu_fallback_tasks"] == tm["minu_tasks"]
    ocb = oracle.codebook(codebook_bytes)
    hr = [oracle.rolled(T.write_rolled(g))[0] for g in rolled]
    n_pos = 0
    for qi, L in enumerate(lats):
        hl, _ = oracle.latent(ocb, T.write_latent(L))
        rc, sc, parts = oracle.search(ocb, hl, hr, tie_mode=1, want_parts=True)
        got = np.concatenate([res["parts"][qi], res["scores"][qi][:, None]], axis=1)
        diff = got.view(np.uint32) != parts.view(np.uint32)
        assert not diff.any(), (qi, np.argwhere(diff)[:4], got[diff][:4], parts[diff][:4])
        assert all(sc[g] > 0 for g in mates[qi])
        n_pos += int((sc > 0).sum())
    assert n_pos >= 2 * len(lats)


def test_candidate_shape_classes_lists_match_oracle_traces(codebook_bytes, cb, oracle):
    """The candidate list (S3: members, order, similarity bits) of pairs in every shape class of k_minu_cands_rt — small (<= 64 x 128), medium (<= 16 384 similarities),
    large (<= 512 rolled minutiae, <= 38 912 similarities incl. the padding column), at the classes' own borders (rt_max_rows) — and just beyond them (any-shape kernel), against the oracle's stage-0 trace; then the same pairs'
    scores through the generic kernel alone (option minu_generic) must be the same bits.  matcher.cpp:440-488."""
    rng = np.random.default_rng(314)
    # (latent minutiae, rolled minutiae) at the class borders of rt_max_rows(): 64 x 128 | 65 x 128, 64 x 129 | 128 x 128 (S = 2: 16 640 / 129 = 128) | 129 x 128 |
    # 64 x 256 (S = 2: two row phases) | 65 x 256 (S = 4) | 128 x 256 (the last shape whose keys all stay in registers) | 129 x 256, 151 x 256 (S = 4 with a second key block;
    # 38 912 / 257 = 151) | 152 x 256 (any-shape kernel) | 256 x 128 | 256 x 150, 200 x 190 (second key block, six / five row phases) | 60 x 300, 97 x 400, 75 x 512 (S = 4 with more column
    # tiles than waves; 38 912 / 401 = 97, / 513 = 75) | 98 x 400, 40 x 513, 257 x 100 (any-shape kernel) | 30 x 17 (tiny)
    shapes = [(64, 128), (65, 128), (64, 129), (128, 128), (129, 128), (64, 256), (65, 256), (128, 256), (129, 256), (151, 256), (152, 256), (256, 128), (256, 150), (200, 190),
              (60, 300), (97, 400), (75, 512), (98, 400), (40, 513), (257, 100), (30, 17)]
    lat_sizes = sorted({s[0] for s in shapes})
    lats = {}
    for nl in lat_sizes:
        lats[nl] = S.make_latent(rng, n_tex_lo=100, n_tex_hi=120, n_minu_lo=nl, n_minu_hi=nl)
    gal, pairs = [], []
    for nl, nr in shapes:
        gal.append(S.make_mate(rng, cb, lats[nl], frac=0.6, n_minu=nr, n_tex=300)); pairs.append((nl, len(gal) - 1))
        gal.append(S.make_rolled(rng, cb, n_minu=nr, n_tex=300)); pairs.append((nl, len(gal) - 1))
    m = M.Matcher(codebook_bytes, taps=True)
    m.gallery_add(gal); m.gallery_commit(0)
    ocb = oracle.codebook(codebook_bytes)
    hr = [oracle.rolled(T.write_rolled(g))[0] for g in gal]
    n_lists = 0
    for nl, gi in pairs:
        hl, _ = oracle.latent(ocb, T.write_latent(lats[nl]))
        for which in range(3):
            want = oracle.trace(ocb, hl, hr[gi], which=which, stage=0, tie_mode=1)
            got = m.debug_stage_list(lats[nl], gi, which, 0)
            assert (want is None) == (got is None), (nl, gi, which)
            if want is None: continue
            ws, wl, wr = want
            assert np.array_equal(got[1], wl) and np.array_equal(got[2], wr), (nl, gal[gi].minu[0].n, which)
            assert np.array_equal(got[0].view(np.uint32), ws.view(np.uint32)), (nl, gal[gi].minu[0].n, which)
            n_lists += 1
    assert n_lists == 3 * len(pairs)
    L = [lats[nl] for nl in lat_sizes]
    fast = m.search(L, k=0, want_parts=True); tm = m.timing()
    assert tm["minu_tasks_small"] > 0 and tm["minu_tasks_medium"] > 0 and tm["minu_tasks_large"] > 0 and tm["minu_fallback_tasks"] > 0, tm
    m.set_option("minu_generic", 1)
    gen = m.search(L, k=0, want_parts=True)
    assert np.array_equal(fast["parts"].view(np.uint32), gen["parts"].view(np.uint32))
    assert m.get_option("minu_fast_max_latent") == 256 and m.get_option("minu_fast_max_rolled") == 512 and m.get_option("minu_fast_max_cells") == 38912
    m.close()


def test_structured_templates_against_oracle(codebook_bytes, cb, oracle):
    """A 200-pair slice of the structured sweep (tools/parity_sweep.py with AFIS_SWEEP_WORKLOAD=structured; host/synth_structured.py): rolled texture points on the extractor's
    16-px grid inside a foreground blob (unique coordinates, scan order: extraction_rolled.py:112-128), orientations from a smooth ridge-flow field (:125), descriptors near a
    shared manifold that are PQ-encoded afterwards — neighbouring points share most of their 16 codes and about 10 % / 30 % of a template's points carry a code vector that
    occurs twice in it (the case `std::max_element`'s first-maximum rule exists for, matcher.cpp:730) — latents with two orientations per grid point (extraction_latent.py:204-205).
    Unlike the i.i.d. templates most NON-mates score above zero here (smooth fields survive the angle tests of matcher.cpp:1503-1540), so every pair exercises the graph stages.
    Every per-part and fused score bit for bit; the candidate lists of a few pairs after every stage; the crowded-threshold-bin route of the candidate kernel must have kept the
    tasks away from the any-shape kernel."""
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
    n_pos = n_pairs = n_fill = 0
    for dup, n_lat, n_gal, idw, sg in ((10, 4, 30, 0.3, SS.DUP_SIGMA[10]), (30, 2, 40, 1.0, 0.0057)):      # identity weight 1.0 (0.0057 = its noise level for 30 % repeats): a tenth of the minutiae lists has fewer than 120 positive similarities (the zero-fill route of the candidate kernel)
        SS.IDENTITY_WEIGHT = idw
        rng = np.random.default_rng(600 + dup)
        lats = [SS.make_structured_latent(rng, sigma=sg) for _ in range(n_lat)]
        gal = [SS.make_structured_mate(rng, cb, L, frac=f, sigma=sg) for L in lats for f in (0.8, 0.3)]
        while len(gal) < n_gal: gal.append(SS.make_structured_rolled(rng, cb, sigma=sg))
        off = np.concatenate([[0], np.cumsum([g.tex[0].n for g in gal])])
        share = SS.dup_share(np.concatenate([g.tex[0].codes for g in gal]), off)
        assert 0.4 * dup / 100 <= share <= 2.0 * dup / 100, share
        m = M.Matcher(codebook_bytes, taps=True)
        m.gallery_add(gal); m.gallery_commit(0)
        res = m.search(lats, k=0, want_parts=True); tm = m.timing()
        assert tm["minu_fallback_tasks"] <= 0.02 * tm["minu_tasks"], tm
        ocb = oracle.codebook(codebook_bytes)
        hl, hr = cases.to_orc(oracle, ocb, lats, gal)
        for qi in range(n_lat):
            rc, sc, parts = oracle.search(ocb, hl[qi], hr, tie_mode=1, want_parts=True)
            got = np.concatenate([res["parts"][qi], res["scores"][qi][:, None]], axis=1)
            diff = got.view(np.uint32) != parts.view(np.uint32)
            assert not diff.any(), (dup, qi, np.argwhere(diff)[:4], got[diff][:4], parts[diff][:4])
            n_pos += int((sc > 0).sum()); n_pairs += len(gal)
            assert sc[2 * qi] > 50 and int(np.argmax(sc)) == 2 * qi
        for qi, gi in ((0, 0), (0, n_gal - 1), (1, n_gal - 2)):               # a mate and two non-mates: the lists after every stage
            for which in range(4):
                for stage in range(3):
                    want = oracle.trace(ocb, hl[qi], hr[gi], which=which, stage=stage, tie_mode=1)
                    gotl = m.debug_stage_list(lats[qi], gi, which, stage)
                    assert (want is None) == (gotl is None)
                    if want is None: continue
                    assert np.array_equal(gotl[1], want[1]) and np.array_equal(gotl[2], want[2]), (dup, qi, gi, which, stage)
                    assert np.array_equal(gotl[0].view(np.uint32), want[0].view(np.uint32)), (dup, qi, gi, which, stage)
        for L in lats:                                                     # lists short of 120 positive similarities exist in the second set (numpy restatement of S1's sign)
            for s_ in (26, 2, 11):
                for R in gal: n_fill += int(((L.minu[s_].des @ R.minu[0].des.T) > 0).sum() < 120 and L.minu[s_].n * R.minu[0].n >= 512)
        m.close()
    SS.IDENTITY_WEIGHT = 0.3
    assert n_fill >= 5, n_fill
    assert n_pairs == 200 and n_pos >= 100, (n_pairs, n_pos)                   # most non-mates score above zero


def test_candidate_ties_in_the_reference_sort_order(codebook_bytes, cb, oracle, tmp_path):
    """Option s3_tie_order 1: where candidate norms tie, the list of 120 is the one libstdc++'s std::sort leaves (matcher.cpp:473-476 sorts the nL x nR indices with a non-strict
    comparator) — csrc/stdsort_order.h run by the any-shape candidate kernel (arrays in LDS up to 8192 similarities, in global scratch beyond) — instead of ascending element index.  It matters for lists with fewer than 120 POSITIVE similarities
    (the zeros that fill the list are all tied): tiny latent templates, and prints whose descriptors point away from each other (structured templates at identity weight 1.0).
    Equal POSITIVE norms inside a full list (two rolled minutiae with the same descriptor) are ordered the same way: the fast kernel detects them while ranking and hands the list over.
    The oracle's tie mode 4 takes std::sort at S3 and the stable order elsewhere: per-part and fused scores and the S3 lists bit for bit; and the test is not vacuous — the
    default order (tie mode 1) gives other scores on some of the pairs."""
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
    rng = np.random.default_rng(905)
    SS.IDENTITY_WEIGHT = 1.0
    try:
        lats = [SS.make_structured_latent(rng, sigma=0.0095, n_tex_lo=200, n_tex_hi=260) for _ in range(3)]
        lats.append(SS.make_structured_latent(rng, sigma=0.0095, n_tex_lo=200, n_tex_hi=260, n_minu_lo=3, n_minu_hi=6))          # tiny lists: nL x nR / 2 < 120 against small rolled prints
        gal = [SS.make_structured_mate(rng, cb, L, frac=0.6, sigma=0.0095, n_minu=int(rng.integers(40, 120)), n_tex=320) for L in lats[:3]]
        while len(gal) < 60: gal.append(SS.make_structured_rolled(rng, cb, sigma=0.0095, n_minu=int(rng.integers(20, 128)), n_tex=300))
        # pairs BEYOND 8192 similarities with short lists (the arrays of the restatement sit in global scratch there, 32-bit indices): rolled prints of 400 minutiae whose
        # descriptors point away from latent 0's — every similarity clamps to zero — except two or three rows copied from it (2 nL or 3 nL positive similarities, fewer than 120)
        n_big = 0
        for k_copy in (2, 3, 0):
            R = SS.make_structured_rolled(rng, cb, sigma=0.0095, n_minu=400, n_tex=300)
            src = np.concatenate([lats[0].minu[s_].des for s_ in (26, 2, 11)])
            u = src.mean(axis=0); u /= np.linalg.norm(u)
            des = -u[None, :] * 1.2 + 0.02 * rng.standard_normal(R.minu[0].des.shape)
            des[rng.choice(400, k_copy, replace=False)] = src[rng.choice(len(src), k_copy, replace=False)]
            R.minu[0].des[:] = des.astype(np.float32)
            gal.append(R)
        # equal POSITIVE norms inside full lists (met once in 180 000 lists of the sweeps): rolled prints of 100 minutiae, 60 of them near latent 0's descriptors, two pairs of them identical
        # (same similarities, same column sums: the norms of (i, 4) and (i, 11) are the same float for every latent row i)
        for _ in range(3):
            R = SS.make_structured_rolled(rng, cb, sigma=0.0095, n_minu=100, n_tex=300)
            d = R.minu[0].des
            d[:60] = (src[rng.choice(len(src), 60)] + 0.3 * rng.standard_normal((60, d.shape[1]))).astype(np.float32)
            d[11] = d[4]; d[37] = d[20]
            gal.append(R)
    finally:
        SS.IDENTITY_WEIGHT = 0.3
    for s_ in (26, 2, 11):
        for R in gal[63:]: assert int(((lats[0].minu[s_].des @ R.minu[0].des.T) > 0).sum()) >= 1000       # full lists: not the zero-fill mechanism
        for R in gal[60:63]:
            npos = int(((lats[0].minu[s_].des @ R.minu[0].des.T) > 0).sum())
            n_big += int(lats[0].minu[s_].n * 400 > 8192 and npos < 120)
    assert n_big >= 6, n_big
    m = M.Matcher(codebook_bytes, taps=True)
    m.gallery_add(gal); m.gallery_commit(0)
    base = m.search(lats, k=0, want_parts=True)
    m.set_option("s3_tie_order", 1)
    assert m.get_option("s3_tie_order") == 1
    res = m.search(lats, k=0, want_parts=True); tm = m.timing()
    assert tm["minu_fallback_tasks"] > 0
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)
    n_moved = n_short = n_big_moved = 0
    for qi in range(len(lats)):
        rc, sc4, p4 = oracle.search(ocb, hl[qi], hr, tie_mode=4, want_parts=True)
        rc, sc1, p1 = oracle.search(ocb, hl[qi], hr, tie_mode=1, want_parts=True)
        got = np.concatenate([res["parts"][qi], res["scores"][qi][:, None]], axis=1)
        diff = got.view(np.uint32) != p4.view(np.uint32)
        assert not diff.any(), (qi, np.argwhere(diff)[:4], got[diff][:4], p4[diff][:4])
        gb = np.concatenate([base["parts"][qi], base["scores"][qi][:, None]], axis=1)
        assert np.array_equal(gb.view(np.uint32), p1.view(np.uint32))                      # the default stays the ascending-index order
        n_moved += int((p4.view(np.uint32) != p1.view(np.uint32)).any(axis=1).sum())
        for s_ in (26, 2, 11):
            for R in gal: n_short += int(((lats[qi].minu[s_].des @ R.minu[0].des.T) > 0).sum() < min(120, lats[qi].minu[s_].n * R.minu[0].n))
    assert n_short >= 20 and n_moved >= 3, (n_short, n_moved)
    for qi, gi in ((0, 5), (3, 7), (3, 20), (1, 33), (0, 60), (0, 61), (0, 62), (0, 63), (0, 64), (0, 65)):   # the S3 lists themselves
        for which in (1, 2, 3):
            want = oracle.trace(ocb, hl[qi], hr[gi], which=which, stage=0, tie_mode=4)
            gotl = m.debug_stage_list(lats[qi], gi, which, 0)
            assert (want is None) == (gotl is None)
            if want is None: continue
            assert np.array_equal(gotl[1], want[1]) and np.array_equal(gotl[2], want[2]), (qi, gi, which)
            if gi >= 60:                                                                    # ... and on the large pairs and the lists with equal positive norms it is not the ascending-index list
                w1 = oracle.trace(ocb, hl[qi], hr[gi], which=which, stage=0, tie_mode=1)
                n_big_moved += int(not (np.array_equal(w1[1], want[1]) and np.array_equal(w1[2], want[2])))
    assert n_big_moved >= 10, n_big_moved
    with pytest.raises(M.AfisError):
        m.set_option("s3_tie_order", 2)
    # ref_tie_order 2: the greedy selections of S8 and S9 (matcher.cpp:1301 / :1423 / :1590) walk equal SCORES in std::sort's order too (graph.hip::sort_scores) — the oracle's tie
    # mode 9.  That is where the MATES are: a mated pair keeps dozens of correspondences whose S9 scores (boolean H, uniform start vector) tie exactly.
    m.set_option("ref_tie_order", 2)
    assert m.get_option("ref_tie_order") == 2 and m.get_option("s3_tie_order") == 1
    res = m.search(lats, k=0, want_parts=True)
    n_moved89 = 0
    for qi in range(len(lats)):
        rc, sc9, p9 = oracle.search(ocb, hl[qi], hr, tie_mode=9, want_parts=True)
        rc, sc4, p4 = oracle.search(ocb, hl[qi], hr, tie_mode=4, want_parts=True)
        got = np.concatenate([res["parts"][qi], res["scores"][qi][:, None]], axis=1)
        diff = got.view(np.uint32) != p9.view(np.uint32)
        assert not diff.any(), (qi, np.argwhere(diff)[:4], got[diff][:4], p9[diff][:4])
        moved = (p9.view(np.uint32) != p4.view(np.uint32)).any(axis=1)
        n_moved89 += int(moved.sum())
        if qi < 3: assert moved[qi], qi                                                     # the planted mate of latent qi is gallery template qi
    assert n_moved89 >= 3, n_moved89
    for qi, gi in ((0, 0), (1, 1), (2, 2)):                                                # the survivors of S8 and of S9 on the mates
        for which in (0, 1, 2, 3):
            for stage in (1, 2):
                want = oracle.trace(ocb, hl[qi], hr[gi], which=which, stage=stage, tie_mode=9)
                gotl = m.debug_stage_list(lats[qi], gi, which, stage)
                assert (want is None) == (gotl is None)
                if want is None: continue
                assert np.array_equal(gotl[1], want[1]) and np.array_equal(gotl[2], want[2]), (qi, gi, which, stage)
    # the all-templates mode (matcher.cpp:339-374) runs the same kernels: latent 1 against its mate, every template, tie mode 9
    qs_, rs_, sc_all = m.One2One_matching_all_templates(lats[1])
    width = len(lats[1].minu) + len(lats[1].tex)
    rc, want_all = oracle.all_templates(ocb, hl[1], hr[1], width, tie_mode=9)
    rc, base_all = oracle.all_templates(ocb, hl[1], hr[1], width, tie_mode=4)
    assert np.array_equal(sc_all[1].view(np.uint32), want_all.view(np.uint32)) and not np.array_equal(want_all.view(np.uint32), base_all.view(np.uint32))
    m.set_option("s3_tie_order", 1)
    assert m.get_option("ref_tie_order") == 1
    with pytest.raises(M.AfisError):
        m.set_option("ref_tie_order", 3)
    m.close()
    # the CLI: `match -ldir ... -tie 2` writes the scores of tie mode 9, without the flag those of tie mode 1 (the mate of latent 1 differs in the first decimal)
    import subprocess
    exe = os.path.join(os.path.dirname(M.LIB_PATH), "match")
    for d in ("gal", "lat", "o1", "o2", "work"): (tmp_path / d).mkdir()
    for j in range(6): (tmp_path / "gal" / f"R{j:03d}.dat").write_bytes(T.write_rolled(gal[j]))
    (tmp_path / "lat" / "L1.dat").write_bytes(T.write_latent(lats[1]))
    cbp = tmp_path / "cb.dat"; cbp.write_bytes(codebook_bytes)
    rc, s9, _ = oracle.search(ocb, hl[1], hr[:6], tie_mode=9, want_parts=True)
    rc, s1, _ = oracle.search(ocb, hl[1], hr[:6], tie_mode=1, want_parts=True)
    assert ["%.3f" % v for v in s9] != ["%.3f" % v for v in s1]
    for flags, out_dir, want in ((["-tie", "2"], "o2", s9), ([], "o1", s1)):
        o = subprocess.run([exe, "-ldir", str(tmp_path / "lat"), "-g", str(tmp_path / "gal"), "-c", str(cbp), "-s", str(tmp_path / out_dir) + "/"] + flags, capture_output=True, text=True, cwd=tmp_path / "work")
        assert o.returncode == 0, o.stderr
        lines = (tmp_path / out_dir / "L1.csv").read_text().splitlines()
        got = {int(os.path.basename(l.rsplit(",", 1)[0].strip('"'))[1:4]): l.rsplit(",", 1)[1] for l in lines if l.startswith('"')}      # (directory order is the file system's)
        assert got == {j: "%.3f" % v for j, v in enumerate(want)}, (flags, got, want)
    o = subprocess.run([exe, "-ldir", str(tmp_path / "lat"), "-g", str(tmp_path / "gal"), "-c", str(cbp), "-s", str(tmp_path / "o1") + "/", "-tie", "7"], capture_output=True, text=True, cwd=tmp_path / "work")
    assert o.returncode == 2 and "ref_tie_order" in o.stderr
    # the rank list of -l (matcher.cpp:306-309: std::sort of the gallery indices by score): a gallery of 29 prints of which 24 score zero against latent 3 — with -tie the tied zeros of the
    # top 24 come out in std::sort's order over the directory listing (the oracle's orc_rank_list calls std::sort on the same array), without it by ascending position
    rc, s9 = oracle.search(ocb, hl[3], hr, tie_mode=9)
    zero = [j for j in range(len(gal)) if s9[j] == 0][:24]; pos = [j for j in range(len(gal)) if s9[j] > 0][:5]
    assert len(zero) == 24 and len(pos) == 5
    for d in ("gal3", "lat3", "o3", "o4", "o5"): (tmp_path / d).mkdir()
    for j in zero + pos: (tmp_path / "gal3" / f"R{j:03d}.dat").write_bytes(T.write_rolled(gal[j]))
    (tmp_path / "lat3" / "L3.dat").write_bytes(T.write_latent(lats[3]))
    base = [exe, "-g", str(tmp_path / "gal3"), "-c", str(cbp)]
    o = subprocess.run(base + ["-ldir", str(tmp_path / "lat3"), "-s", str(tmp_path / "o3") + "/", "-tie", "2"], capture_output=True, text=True, cwd=tmp_path / "work")
    assert o.returncode == 0, o.stderr
    listing = [int(os.path.basename(l.rsplit(",", 1)[0].strip('"'))[1:4]) for l in (tmp_path / "o3" / "L3.csv").read_text().splitlines() if l.startswith('"')]   # the directory order the CLI saw
    assert sorted(listing) == sorted(zero + pos)
    col = np.array([s9[j] for j in listing], np.float32)
    rc, s1 = oracle.search(ocb, hl[3], hr, tie_mode=1)                                      # (without the flag the SCORES are tie mode 1's: a tiny latent's differ)
    col1 = np.array([s1[j] for j in listing], np.float32)
    want_ref = [listing[i] for i in oracle.rank_list(col, True)[:24]]; want_idx = [listing[i] for i in oracle.rank_list(col1, False)[:24]]
    assert want_ref != [listing[i] for i in oracle.rank_list(col, False)[:24]]
    for flags, out_dir, want in ((["-tie", "2"], "o4", want_ref), ([], "o5", want_idx)):
        o = subprocess.run(base + ["-l", str(tmp_path / "lat3" / "L3.dat"), "-s", str(tmp_path / out_dir) + "/"] + flags, capture_output=True, text=True, cwd=tmp_path / "work")
        assert o.returncode == 0, o.stderr
        rows = (tmp_path / out_dir / "L3.csv").read_text().splitlines()[1:]
        got = [int(os.path.basename(r.split('"')[1])[1:4]) for r in rows]
        assert got == want, (flags, got, want)
    # host/matcher.py's One2List_matching under the option: the file the CLI writes with -tie 2 (afis_rank_list behind both)
    (tmp_path / "o6").mkdir()
    m5 = M.Matcher(codebook_bytes); m5.load_gallery_dir(str(tmp_path / "gal3")); m5.set_option("ref_tie_order", 2)
    assert m5.One2List_matching(str(tmp_path / "lat3" / "L3.dat"), str(tmp_path / "o6") + "/") == 0
    assert (tmp_path / "o6" / "L3.csv").read_bytes() == (tmp_path / "o4" / "L3.csv").read_bytes()
    m5.close()


def test_texture_top200_with_row_maxima_of_both_signs(codebook_bytes, cb, oracle):
    """S7 on a latent of 201 .. 256 texture rows whose 200 best row maxima have BOTH signs and reach 2.0 (prints whose descriptors point away from each other: structured templates
    at identity weight 1.0).  When exactly 200 rows pass the recomputation's bound test and some of their maxima are negative but above -2, the list kernel's bit-by-bit search for the
    200th key stops at the coarse prefix 0x40000000 (count == 200 at bit 30), and a maximum >= 2.0 has an ordered key >= 0xC0000000: 2^31 or more above it.  Rounds 3-5 took that
    difference as an int when they binned the 200 keys for ranking — a negative number, a zero shift, bins far out of range, counters scattered over the list's LDS, ranking loops of 2^31
    trips: 72 s for one search (the results stayed right: the tie fallback re-ranked).  Scores against the oracle, the texture stage's time, and a host emulation of the search that
    shows such lists are in the set."""
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
    rng = np.random.default_rng(905)
    SS.IDENTITY_WEIGHT = 1.0
    try:
        lats = [SS.make_structured_latent(rng, sigma=0.0095, n_tex_lo=200, n_tex_hi=260) for _ in range(2)]
        gal = [SS.make_structured_rolled(rng, cb, sigma=0.0095, n_minu=int(rng.integers(20, 128)), n_tex=300) for _ in range(40)]
    finally:
        SS.IDENTITY_WEIGHT = 0.3
    assert all(200 < L.tex[0].n <= 256 for L in lats)
    m = M.Matcher(codebook_bytes, taps=True)
    m.gallery_add(gal); m.gallery_commit(0)
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats, gal)

    def ordered_key(v):
        u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
        return np.where(u & 0x80000000, (~u) & 0xffffffff, u | 0x80000000)

    def search_T(keys):                                                        # graph.hip::k_graph_texture: the 200th largest key, bit by bit, stopping early at an exact count
        t = 0
        for bit in range(31, -1, -1):
            c = int((keys >= (t | (1 << bit))).sum())
            if c >= 200:
                t |= 1 << bit
                if c == 200: break
        return t

    wide = 0
    for qi, L in enumerate(lats):
        res = m.search([L], k=0, want_parts=True); tm = m.timing()
        assert tm["tex_tail_ms"] < 500.0, tm                                   # (72 000 ms before the fix)
        rc, sc, parts = oracle.search(ocb, hl[qi], hr, tie_mode=1, want_parts=True)
        got = np.concatenate([res["parts"][0], res["scores"][0][:, None]], axis=1)
        assert np.array_equal(got.view(np.uint32), parts.view(np.uint32))
        for gi in range(len(gal)):
            val, _ = oracle.texture_rowmax(ocb, hl[qi], hr[gi])
            keys = np.sort(ordered_key(val))[-200:]                            # (exactly the 200 best rows active: the common case for a 224-row latent)
            wide += int(int(keys.max()) - search_T(keys) >= 2 ** 31)
    assert wide >= 3, wide                                                     # lists whose largest key lies 2^31 or more above the search's threshold exist in this set
    m.close()
