"""CPU tests of the parity checker itself: oracle vs committed golden vectors, oracle vs the reference's own header
(oracle/_ref, built from /root/reference/matching/include.h), and the reference's documented edge-case rules."""
import hashlib
import importlib
import os

import numpy as np
import pytest

import cases

T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_pairs.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def handles(gold, oracle, codebook_bytes):
    ocb = oracle.codebook(codebook_bytes)
    hl = [oracle.latent(ocb, gold[f"latent_{i}"].tobytes())[0] for i in range(2)]
    hr = [oracle.rolled(gold[f"rolled_{j}"].tobytes())[0] for j in range(12)]
    return ocb, hl, hr


def test_golden_scores_bit_exact(gold, oracle, handles):
    ocb, hl, hr = handles
    for tm in (0, 1):
        for i, h in enumerate(hl):
            rc, sc, parts = oracle.search(ocb, h, hr, tie_mode=tm, want_parts=True)
            assert rc == 0
            assert np.array_equal(parts.view(np.uint32), gold["parts"][tm, i].view(np.uint32)), (tm, i)
            assert np.array_equal(sc, parts[:, 4])
    # the planted mates score, in order of overlap; tie modes agree to rounding of the final sums
    p = gold["parts"]
    assert p[1, 0, 0, 4] > p[1, 0, 1, 4] > p[1, 0, 2, 4] > 10 and p[1, 1, 3, 4] > p[1, 1, 4, 4] > p[1, 1, 5, 4] > 10
    # tie_mode 0 (libstdc++ std::sort, what the reference executes) and tie_mode 1 (equal keys by ascending index, what the HIP path
    # implements) visit equal keys in different orders; a tied correspondence can then be swapped for another or dropped.  The
    # reference leaves this order unspecified; pairs that differ are counted, not hidden.
    rel = np.abs(p[0, ..., 4] - p[1, ..., 4]) / np.maximum(1.0, np.abs(p[1, ..., 4]))
    assert (rel > 1e-3).sum() <= 1 and (rel > 1e-3).mean() < 0.05

def test_golden_lut_and_rowmax_hashes(gold, oracle, handles):
    ocb, hl, hr = handles
    for i, h in enumerate(hl):
        assert hashlib.sha256(oracle.lut(h, 0).tobytes()).hexdigest() == str(gold["lut_sha256"][i])
    k = 0
    for h in hl:
        for r in hr:
            v, a = oracle.texture_rowmax(ocb, h, r)
            assert hashlib.sha256(v.tobytes() + a.astype(np.int32).tobytes()).hexdigest() == str(gold["rowmax_sha256"][k])
            k += 1


def test_lut_pinned_by_reference_header(oracle, codebook_bytes):
    """S4: the oracle's LUT equals LatentTextureTemplate::compute_dist_to_codewords compiled from the reference's include.h."""
    from oracle_lib import RefHarness
    try:
        ref = RefHarness()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libafis_ref.so not built (needs /root/reference)")
    cb = T.Codebook.from_bytes(codebook_bytes)
    ocb = oracle.codebook(codebook_bytes)
    rng = np.random.default_rng(5)
    des = (rng.standard_normal((37, 96)) * 0.2).astype(np.float32)
    des[3] = np.concatenate([cb.words[m, (7 * m) % 256] for m in range(16)])     # exact codewords: zero distances
    a = oracle.build_lut(ocb, des); b = ref.build_lut(des, cb.words)
    assert a.shape == b.shape == (37, 16, 256)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[3, 5, 35] == 0.0
    assert abs(ref.lib.ref_pi() - 3.1415926) == 0.0                               # include.h:22


def test_rolled_code_extraction_pinned_by_reference_header(oracle):
    """T1: the PQ codes the reference keeps are the first n*des_len BYTES of the buffer it read as floats (include.h:401-406)."""
    from oracle_lib import RefHarness
    try:
        ref = RefHarness()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libafis_ref.so not built (needs /root/reference)")
    rng = np.random.default_rng(6)
    n = 23
    t = T.FPTemplate(minu=[T.MinutiaeTemplate(np.arange(5, dtype=np.int16), np.arange(5, dtype=np.int16), np.zeros(5, np.float32), np.ones((5, 96), np.float32))],
                     tex=[T.TextureTemplate(rng.integers(0, 45, n).astype(np.int16), rng.integers(0, 47, n).astype(np.int16),
                                            rng.uniform(-1, 1, n).astype(np.float32), codes=rng.integers(0, 256, (n, 16)).astype(np.uint8))])
    buf = T.write_rolled(t)
    h, rc = oracle.rolled(buf)
    assert rc == 0
    x = t.tex[0]
    codes, xo, yo, oo = ref.rolled_texture(x.x, x.y, x.ori, 16, x.codes.tobytes())
    got = np.ctypeslib.as_array(oracle.lib.orc_rolled_codes(h, 0), shape=(n, 16))
    assert np.array_equal(got, codes) and np.array_equal(codes, x.codes)
    assert np.array_equal(xo, x.x.astype(np.int32)) and np.array_equal(oo, x.ori)


def test_table_dist_is_a_correctly_rounded_sqrt(oracle, codebook_bytes):
    """matcher.cpp:45-56: table_dist[i*50+j] = (float)sqrt((16i)^2+(16j)^2).  The HIP path recomputes it as the correctly rounded
    fp32 sqrt of the exact integer 256*(i^2+j^2); both must agree for all 2500 entries."""
    ocb = oracle.codebook(codebook_bytes)
    tab = np.ctypeslib.as_array(oracle.lib.orc_codebook_table_dist(ocb), shape=(50, 50))
    i, j = np.meshgrid(np.arange(50), np.arange(50), indexing="ij")
    s = np.sqrt((256 * (i * i + j * j)).astype(np.float32))
    assert np.array_equal(tab.view(np.uint32), s.view(np.uint32))


def test_fusion_and_status_rules(oracle, codebook_bytes):
    cb = T.Codebook.from_bytes(codebook_bytes)
    ocb = oracle.codebook(codebook_bytes)
    base, variants = cases.edge_latents(cb)
    rng = np.random.default_rng(11)
    mate = oracle.rolled(T.write_rolled(S.make_mate(rng, cb, base, frac=0.8, n_tex=400)))[0]
    res = {}
    for name, L in variants.items():
        if not L.minu:
            continue
        h, _ = oracle.latent(ocb, T.write_latent(L))
        res[name] = oracle.pair(ocb, h, mate, 1)
    rc, p = res["full28"]
    assert rc == 0 and p[3] > 0 and np.isclose(p[4], (p[0] + p[1] + p[2]) + 0.3 * p[3], rtol=1e-6)
    rc, p = res["minu27_tex"]                     # texture lands in score[27]; score[28] is out of range -> reads as 0 here
    assert rc == 0 and np.isclose(p[4], p[0] + p[1] + p[2], rtol=1e-6)
    rc, p = res["minu29_tex"]                     # score[28] is a (zero) minutiae slot
    assert rc == 0 and np.isclose(p[4], p[0] + p[1] + p[2], rtol=1e-6)
    rc, p = res["minu12_tex"]                     # only selected templates 2 and 11 exist: score[0] stays 0
    assert rc == 0 and p[0] == 0 and p[1] > 0 and p[2] > 0
    rc, p = res["minu2_tex"]                      # texture lands in score[2] with weight 1
    assert rc == 0 and np.isclose(p[4], p[3], rtol=1e-6) and p[3] > 0
    assert res["minu26_notex"][0] == 1            # latent "empty" (matcher.cpp:383-386)
    assert res["minu28_notex"][0] == 0 and res["minu28_notex"][1][3] == 0
    # rolled side: a file of <= 10 bytes is an empty template -> status 2 (matcher.cpp:899-902, :388-391)
    h, _ = oracle.latent(ocb, T.write_latent(base))
    e, rc_load = oracle.rolled(b"\x01\x00" * 5)
    assert rc_load == 1 and oracle.pair(ocb, h, e, 1)[0] == 2


def test_zero_minutiae_templates_are_dropped_and_indices_shift(oracle, codebook_bytes):
    """matcher.cpp:835-836: a minutiae template with n <= 0 is skipped, so later templates move down one index."""
    cb = T.Codebook.from_bytes(codebook_bytes)
    ocb = oracle.codebook(codebook_bytes)
    base, _ = cases.edge_latents(cb)
    holey = T.FPTemplate(minu=list(base.minu), tex=list(base.tex))
    e = T.MinutiaeTemplate(np.zeros(0, np.int16), np.zeros(0, np.int16), np.zeros(0, np.float32), np.zeros((0, 96), np.float32))
    holey.minu = holey.minu[:5] + [e] + holey.minu[5:]             # 29 entries on disk, 28 after the drop
    h, rc = oracle.latent(ocb, T.write_latent(holey))
    assert rc == 0 and oracle.counts(h) == (28, 1)
    rc2, t = T.read_latent(T.write_latent(holey))
    assert rc2 == 0 and len(t.minu) == 28


def test_pq_encoder_pinned_to_scipy_golden(oracle, codebook_bytes, templates_mod):
    """the oracle's encoder against codes produced by scipy.cluster.vq.vq — the routine the reference's
    TrainedPQEncoder.encode_multi calls (descriptor_PQ.py:25-26) — on the reference's codebook (tests/golden/make_golden_pq.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_pq.npz"))
    ocb = oracle.codebook(codebook_bytes)
    cb = templates_mod.Codebook.from_bytes(codebook_bytes)
    got = oracle.pq_encode(ocb, g["des"])
    assert np.array_equal(got, g["codes"])
    # rows 0..63 are exact codewords (distance 0): decoding the codes gives them back
    dec = np.concatenate([cb.words[m][got[:64, m]] for m in range(16)], axis=1)
    assert np.array_equal(dec, g["des"][:64])
    # the host-side numpy helper used by the synthetic generator agrees as well
    assert np.array_equal(cb.encode(g["des"]), g["codes"])


def test_cli_token_rules_pinned_by_reference_argparser():
    """`match`'s flag parsing against the reference's own matching/argparser.h (compiled into oracle/_ref): first occurrence wins,
    a flag at the end of the line exists but has an empty value, a value may look like a flag."""
    import os, subprocess
    from oracle_lib import RefHarness
    try:
        ref = RefHarness()
        ref.lib.ref_arg
    except (FileNotFoundError, OSError, AttributeError):
        pytest.skip("oracle/_ref/libafis_ref.so not built with argparser.h (needs /root/reference)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "msu-latentafis_amd", "csrc", "match_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "match_selftest"], check=True)
    lines = [["-l", "a.dat", "-g", "gal/", "-s", "out/", "-c", "cb.dat"], ["-g", "gal", "-l"], ["-l", "-g", "x"], ["-s", "1", "-s", "2"],
             [], ["-ldir", "d", "-l", "f"], ["-c"], ["--c", "x", "-c", "y"]]
    for toks in lines:
        for opt in ("-l", "-ldir", "-g", "-s", "-c", "-d"):
            want = ref.arg(["match"] + toks, opt)
            out = subprocess.run([exe, "-selftest-args", opt] + toks, capture_output=True, text=True, check=True).stdout.rstrip("\n")
            assert out == f"exists={int(want[0])} value={want[1]}", (toks, opt, out, want)


def test_config_fallback_pinned_by_reference_json_library(tmp_path):
    """`../afis.config` as main.cpp:41-44 reads it — with the JSON library vendored in the reference tree — against the flat reader
    of the `match` CLI: the reference's own key set, escapes in values, extra whitespace, non-string values next to the keys."""
    import os, subprocess
    from oracle_lib import RefHarness
    try:
        ref = RefHarness()
        ref.lib.ref_config_get
    except (FileNotFoundError, OSError, AttributeError):
        pytest.skip("oracle/_ref/libafis_ref.so not built with json.hpp (needs /root/reference)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "msu-latentafis_amd", "csrc", "match_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "match_selftest"], check=True)
    keys = ["CodebookPath", "ScorePath", "GalleryTemplateDirectory", "LatentTemplateDirectory", "MinuPath", "Absent"]
    texts = [
        '{\n\t"CodebookPath": "/home/x/codebook.dat",\n\t"ScorePath": "/home/x/scores",\n\n\t"GalleryTemplateDirectory": "/g",\n'
        '\t"LatentTemplateDirectory": "/l",\n\t"MinuPath": "None"\n\n}\n',
        '{"ScorePath":"a b/c",   "CodebookPath" :\n "q\\\\r\\"s" , "Depth": 3, "Flag": true, "GalleryTemplateDirectory": "/g/"}',
        '{ "LatentTemplateDirectory": "", "ScorePath": "/s/", "Nested": {"CodebookPath": "inner"}, "List": ["x", "y"] }',
    ]
    for n, text in enumerate(texts):
        f = tmp_path / f"c{n}.config"; f.write_text(text)
        for key in keys:
            rc, want = ref.config_get(str(f), key)
            out = subprocess.run([exe, "-selftest-config", str(f), "-key", key], capture_output=True, text=True, check=True).stdout.rstrip("\n")
            if rc == 1:
                assert out == f"found=1 value={want}", (n, key, out, want)
            elif n < 2:                                  # flat files: absent in one reader = absent in the other
                assert out == "found=0 value=", (n, key, out)


def test_scores_stay_inside_the_tolerance_under_every_accumulation_order(oracle, codebook_bytes):
    """The reference leaves S1's product, S2's sums and the S8 mat-vecs to Eigen (matcher.cpp:443, :455-470, :1279-1289, :1401-1411; version unpinned, not in the tree): the
    oracle and the HIP path fix one order.  SURVEY section 8d's rule — final scores within 1e-3 * max(1, |s|) for >= 99.9 % of the pairs, the same rank 1 — must hold
    against every order an Eigen build could take (oracle sum_order 1..5; tools/order_sweep.py runs 96 000 pairs per order -> profiles/r06_order_sweep*.json).  Here: 1600
    structured pairs (most of them score above zero, unlike the i.i.d. templates), every order against the canonical one."""
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
    cb = T.Codebook.from_bytes(codebook_bytes)
    sg = SS.DUP_SIGMA[10]
    lats = SS.make_structured_latents(41, 2, sigma=sg)
    gal = SS.make_packed_gallery_structured(41, 800, cb, sigma=sg)
    planted = SS.plant_structured_mates(41, gal, cb, lats, G=800, sigma=sg)
    ocb = oracle.codebook(codebook_bytes)
    hr = [oracle.rolled(T.write_rolled(gal.template(g)))[0] for g in range(gal.G)]
    n_pos = n_bits = 0
    n_far = {so: 0 for so in (1, 2, 3, 4, 5)}
    for qi, L in enumerate(lats):
        hl, _ = oracle.latent(ocb, T.write_latent(L))
        _, s0 = oracle.search(ocb, hl, hr, tie_mode=1)
        assert int(np.argmax(s0)) == planted[qi][0][0]
        n_pos += int((s0 > 0).sum())
        for so in (1, 2, 3, 4, 5):
            _, s1 = oracle.search(ocb, hl, hr, tie_mode=1 | (so << 4))
            far = np.abs(s1 - s0) > 1e-3 * np.maximum(1.0, np.abs(s0))
            n_far[so] += int(far.sum())
            assert int(np.argmax(s1)) == int(np.argmax(s0))
            n_bits += int((s1.view(np.uint32) != s0.view(np.uint32)).sum())
    assert all(v <= 1e-3 * 2 * gal.G for v in n_far.values()), n_far     # >= 99.9 % of the 1600 pairs inside the tolerance, for every order (the 96 000-pair sweeps: 99.997 %)
    assert n_pos > 600 and n_bits > 0          # the orders DO differ in the last bits of many scores: the test is not vacuous


def test_rank_list_orders(oracle):
    """orc_rank_list (matcher.cpp:306-309, the rank list of -l): with the stable order it is the lexsort by (score descending, index ascending) that afis_search's top-k and the
    CLI's default produce; with std::sort it is a permutation in non-increasing score order that agrees with the stable one wherever scores are distinct, and IS the stable one up
    to 16 entries (libstdc++ insertion-sorts those) — and is NOT for a longer list with a tied tail (what `match -l -tie` reproduces)."""
    rng = np.random.default_rng(5)
    for n in (1, 3, 16, 17, 66, 1000):
        s = np.zeros(n, np.float32)
        k = max(1, n // 4)
        s[rng.choice(n, k, replace=False)] = rng.random(k).astype(np.float32) * 100
        st = oracle.rank_list(s, False); sd = oracle.rank_list(s, True)
        assert np.array_equal(st, np.lexsort((np.arange(n), -s.astype(np.float64))))
        assert sorted(sd.tolist()) == list(range(n)) and (np.diff(s[sd]) <= 0).all()
        assert np.array_equal(sd[:k], st[:k])                                   # the distinct positive scores lead both lists
        if n <= 16: assert np.array_equal(sd, st)
    s = np.zeros(66, np.float32); s[[5, 9, 40]] = [3, 7, 1]
    assert not np.array_equal(oracle.rank_list(s, True)[:24], oracle.rank_list(s, False)[:24])


def test_where_the_reference_sort_order_moves_scores(oracle, codebook_bytes):
    """The oracle's site-wise tie modes on structured templates (host/synth_structured.py), the CPU side of option ref_tie_order: a MATED pair keeps dozens of correspondences whose
    S9 scores tie exactly, so std::sort at S3 alone (mode 4 = ref_tie_order 1) does not give the reference binary's score (mode 0) for the mates, std::sort at S3 + S8 + S9 (mode 9 =
    ref_tie_order 2) does; and a pair with a tiny latent template (lists filled with tied zero norms) differs already at S3 (mode 4 != mode 1)."""
    SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")
    cb = T.Codebook.from_bytes(codebook_bytes)
    rng = np.random.default_rng(905)
    SS.IDENTITY_WEIGHT = 1.0
    try:
        lats = [SS.make_structured_latent(rng, sigma=0.0095, n_tex_lo=200, n_tex_hi=260) for _ in range(3)]
        tiny = SS.make_structured_latent(rng, sigma=0.0095, n_tex_lo=200, n_tex_hi=260, n_minu_lo=3, n_minu_hi=6)
        gal = [SS.make_structured_mate(rng, cb, L, frac=0.6, sigma=0.0095, n_minu=int(rng.integers(40, 120)), n_tex=320) for L in lats]
        while len(gal) < 40: gal.append(SS.make_structured_rolled(rng, cb, sigma=0.0095, n_minu=int(rng.integers(20, 128)), n_tex=300))
    finally:
        SS.IDENTITY_WEIGHT = 0.3
    ocb = oracle.codebook(codebook_bytes)
    hl, hr = cases.to_orc(oracle, ocb, lats + [tiny], gal)
    bits = lambda a: a.view(np.uint32)
    mates_moved = 0
    for qi in range(3):
        sc = {m: oracle.search(ocb, hl[qi], hr, tie_mode=m)[1] for m in (0, 1, 4, 9)}
        assert np.array_equal(bits(sc[9]), bits(sc[0])), qi                         # S7's order does not matter on this set
        mates_moved += int(bits(sc[4])[qi] != bits(sc[0])[qi])
        rest = np.arange(len(gal)) != qi
        assert (np.abs(sc[4][rest] - sc[0][rest]) <= 1e-3 * np.maximum(1, np.abs(sc[0][rest]))).all()
    assert mates_moved >= 2, mates_moved
    s1 = oracle.search(ocb, hl[3], hr, tie_mode=1)[1]; s4 = oracle.search(ocb, hl[3], hr, tie_mode=4)[1]; s0 = oracle.search(ocb, hl[3], hr, tie_mode=0)[1]
    assert not np.array_equal(bits(s1), bits(s4)) and int((bits(s4) != bits(s0)).sum()) <= int((bits(s1) != bits(s0)).sum())

