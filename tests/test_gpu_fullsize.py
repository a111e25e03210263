"""GPU parity at BASELINE.json's full sizes: configs[1] (1 latent x 10 000 rolled templates) and configs[2] (batch of 100
latents x 100 000 templates, one step), through the C ABI.

At these sizes the oracle cannot score everything in a test's time budget on every host, so each test combines
  * size-independent properties over ALL pairs (planted mates lead every rank list in planting order; the rank list is
    the lexsort (score descending, index ascending) of the score vector; fusion identity on every pair; -1/0 conventions;
    ADC variants give identical bits), with
  * a bit-for-bit oracle sample: every planted mate plus >= 200 random non-mates per sampled query (configs[2]), or
    every pair of the gallery (configs[1]; 10 000 oracle pairs take about a second on the GPU box's host cores).
The oracle runs in tie_mode=1 (equal keys by ascending index = what the HIP path implements) for the bit-exact checks and in
tie_mode=0 (libstdc++ std::sort order = what the reference binary executes, matcher.cpp:476, :741, :1301, :1423, :1590)
for the tolerance statistics SURVEY section 8d states: |d| <= 1e-3*max(1,|s|) for >= 99.9 % of pairs.
"""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
T = importlib.import_module("msu-latentafis_amd.host.templates")
S = importlib.import_module("msu-latentafis_amd.host.synth")
M = importlib.import_module("msu-latentafis_amd.host.matcher")


@pytest.fixture(scope="module")
def cb(codebook_bytes):
    return T.Codebook.from_bytes(codebook_bytes)


def _oracle_rows(oracle, ocb, latent, gal, gidx, tie_mode):
    """[len(gidx), 5] = (s0, s1, s2, tex, fused) of `latent` against gallery templates gidx, all host threads."""
    hl, _ = oracle.latent(ocb, T.write_latent(latent))
    hr = [oracle.rolled(T.write_rolled(gal.template(int(g))))[0] for g in gidx]
    rc, sc, parts = oracle.search(ocb, hl, hr, tie_mode=tie_mode, threads=oracle.lib.orc_num_threads(), want_parts=True)
    assert rc == 0
    for h in hr:
        oracle.lib.orc_rolled_free(h)
    oracle.lib.orc_latent_free(hl)
    return parts


def _got_rows(res, q, gidx):
    return np.concatenate([res["parts"][q][gidx], res["scores"][q][gidx][:, None]], axis=1)


def _check_properties(res, planted, G, k):
    Q = res["scores"].shape[0]
    p = res["parts"]
    fused = ((p[..., 0] + p[..., 1]) + p[..., 2]).astype(np.float64) + p[..., 3].astype(np.float64) * 0.3   # matcher.cpp:188
    assert np.array_equal(fused.astype(np.float32), res["scores"])
    assert (res["scores"] >= 0).all()                                   # no empty rolled template in the synthetic gallery
    ar = np.arange(G)
    for q in range(Q):
        want = [g for g, _ in planted[q]]
        assert list(res["topk_idx"][q][:len(want)]) == want, (q, res["topk_idx"][q][:6], want)
        order = np.lexsort((ar, -res["scores"][q].astype(np.float64)))[:k]                                     # matcher.cpp:306-309 + tie rule
        assert np.array_equal(res["topk_idx"][q], order), q
        assert np.array_equal(res["topk_score"][q], res["scores"][q][order]), q


def test_config1_one_latent_vs_10k(codebook_bytes, cb, oracle):
    """BASELINE.json configs[1]: every one of the 10 000 pairs against the oracle, bit for bit; tie_mode=0 within tolerance."""
    G, seed = 10000, 4101
    lats = S.make_latents(seed, 1)
    gal = S.make_packed_gallery(seed, G, cb)
    planted = S.plant_mates(seed, gal, cb, lats, G=G, n_partial=5)
    m = M.Matcher(codebook_bytes)
    m.gallery_add_packed(gal); m.gallery_commit(0)
    res = m.search(lats, k=24, want_parts=True)
    _check_properties(res, planted, G, 24)
    for v in (1, 6, 7, 8):                                              # other ADC variants (8 = 16-bit bound pass + exact refine), same bits
        m.set_option("adc_variant", v)
        assert np.array_equal(m.search(lats, k=0)["scores"], res["scores"]), v
    m.close()
    ocb = oracle.codebook(codebook_bytes)
    allg = np.arange(G)
    want1 = _oracle_rows(oracle, ocb, lats[0], gal, allg, 1)
    got = _got_rows(res, 0, allg)
    diff = got.view(np.uint32) != want1.view(np.uint32)
    assert not diff.any(), ("first differing pair", int(np.argwhere(diff.any(axis=1))[0, 0]), int(diff.any(axis=1).sum()))
    assert (want1[:, :4] > 0).sum() > 1000                               # the sample is not a wall of zeros
    want0 = _oracle_rows(oracle, ocb, lats[0], gal, allg, 0)
    far = np.abs(got - want0) > 1e-3 * np.maximum(1.0, np.abs(want0))
    assert far.any(axis=1).mean() <= 1e-3, far.any(axis=1).sum()        # >= 99.9 % of pairs within tolerance
    r0 = np.lexsort((allg, -want0[:, 4].astype(np.float64)))[:24]
    n_pos = int((want0[r0, 4] > 0).sum())
    assert n_pos >= 6 and np.array_equal(res["topk_idx"][0][:n_pos], r0[:n_pos])   # rank list over strictly positive scores


@pytest.fixture(scope="module")
def headline(codebook_bytes, cb):
    """BASELINE.json configs[2]: 100 latents x 100 000 templates, the bench workload (same generator, another seed)."""
    G, Q, seed = 100000, 100, 909
    if os.environ.get("AFIS_TEST_SMALL_HEADLINE"):                     # local debugging only
        G, Q = 20000, 16
    lats = S.make_latents(seed, Q)
    gal = S.make_packed_gallery(seed, G, cb)
    planted = S.plant_mates(seed, gal, cb, lats, G=G)
    m = M.Matcher(codebook_bytes)
    m.gallery_add_packed(gal); m.gallery_commit(0)
    res = m.search(lats, k=24, want_parts=True)
    yield lats, gal, planted, m, res
    m.close()


def test_config2_properties_all_pairs(headline):
    lats, gal, planted, m, res = headline
    _check_properties(res, planted, gal.G, 24)
    assert all(res["topk_score"][q][0] > 50 for q in range(len(lats)))
    # a second pass over resident queries gives the same bits (idempotence), without asking for the score matrix
    qh = m.upload_queries(lats)
    r2 = m.search_resident(qh, k=24)
    m.free_queries(qh)
    assert np.array_equal(r2["topk_idx"], res["topk_idx"]) and np.array_equal(r2["topk_score"], res["topk_score"])


def test_config2_oracle_sample_bit_exact(headline, codebook_bytes, oracle):
    lats, gal, planted, m, res = headline
    G, Q = gal.G, len(lats)
    ocb = oracle.codebook(codebook_bytes)
    rng = np.random.default_rng(5)
    n_pairs = n_nz = n_far0 = 0
    for q in range(Q):
        mates = np.array([g for g, _ in planted[q]])
        gidx = mates if q % 12 else np.unique(np.concatenate([mates, rng.integers(0, G, 260), res["topk_idx"][q]]))
        want = _oracle_rows(oracle, ocb, lats[q], gal, gidx, 1)
        got = _got_rows(res, q, gidx)
        diff = got.view(np.uint32) != want.view(np.uint32)
        assert not diff.any(), (q, int(gidx[np.argwhere(diff.any(axis=1))[0, 0]]), got[diff.any(axis=1)][:2], want[diff.any(axis=1)][:2])
        n_pairs += len(gidx); n_nz += int((want[:, :4] > 0).sum())
        if q % 12 == 0:                                                  # the reference's own sort order: tolerance, not bits
            want0 = _oracle_rows(oracle, ocb, lats[q], gal, gidx, 0)
            n_far0 += int((np.abs(got - want0) > 1e-3 * np.maximum(1.0, np.abs(want0))).any(axis=1).sum())
    assert n_pairs >= 9 * 260 + 4 * Q * 0.9 and n_nz > 500
    assert n_far0 <= max(1, n_pairs // 1000), n_far0


def test_config2_variants_same_bits_on_a_query_slice(headline):
    lats, gal, planted, m, res = headline
    sub = [3, 41, 77] if len(lats) > 77 else [1, 2, 3]
    for v, gen in ((0, 0), (7, 1), (8, 0)):                             # plain-layout ADC kernel; generic minutiae candidate kernel; bound + refine
        m.set_option("adc_variant", v); m.set_option("minu_generic", gen)
        r = m.search([lats[i] for i in sub], k=24)
        assert np.array_equal(r["scores"], res["scores"][sub]) and np.array_equal(r["topk_idx"], res["topk_idx"][sub]), (v, gen)
    m.set_option("adc_variant", 7); m.set_option("minu_generic", 0)
