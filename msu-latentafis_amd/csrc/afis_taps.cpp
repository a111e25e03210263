// afis_taps.cpp — the parity taps (include/afis_matcher_taps.h: afis_debug_*): stage intermediates for tests/.  Built ONLY into libafis_hip_test.so; the product
// library exports none of them.
#include "afis_ctx.h"
#include "../../include/afis_matcher_taps.h"

using namespace afis;

extern "C" {

int afis_debug_phase_cycles(afis_ctx* ctx, unsigned long long* out32, int reset)
{
    if (!ctx || !out32) return AFIS_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, read_phase_cycles(out32, reset != 0));
    unsigned long long gph[16];                          // graph.hip phases (only in PHASE_TIMING builds) reported in slots 0..15 + 32.. is not
    HIPCHK(ctx, read_graph_phase_cycles(gph, reset != 0)); // possible with a 32-slot array: they overlay the unused slots 5..15 and 21..25
    for (int i = 0; i < 8; ++i) { out32[5 + i] = gph[i]; out32[21 + i] = gph[8 + i]; }
    return AFIS_OK;
}

// adc_variant 9, after afis_set_option("mf_stats", 1): counters of the selection / recomputation kernel accumulated since the last reset:
// out[0] pairs, [1] latent rows, [2] rows evaluated (may reach the top 200), [3] candidate cells evaluated, [4] rows evaluated over every point,
// [5] rows whose exact maximum lay outside its bounds (self-check, must be 0)
int afis_debug_refine_stats(afis_ctx* ctx, unsigned long long* out8, int reset)
{
    if (!ctx || !out8) return fail(ctx, AFIS_EINVAL, "afis_debug_refine_stats: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!ctx->mf_stats.p) return AFIS_OK;
    HIPCHK(ctx, hipMemcpy(out8, ctx->mf_stats.p, 64, hipMemcpyDeviceToHost));
    if (reset) HIPCHK(ctx, hipMemset(ctx->mf_stats.p, 0, 64));
    return AFIS_OK;
}

int afis_debug_atan2_grid(afis_ctx* ctx, int R, float* out)
{
    if (!ctx || !out || R < 0 || R > 4096) return fail(ctx, AFIS_EINVAL, "afis_debug_atan2_grid: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)(2 * R + 1) * (2 * R + 1);
    DevBuf d;
    HIPCHK(ctx, d.ensure(n * 4));
    hipError_t e = launch_debug_atan2_grid(R, d.as<float>(), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d.p, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d.release();
    if (e != hipSuccess) return fail(ctx, AFIS_EDEVICE, std::string("afis_debug_atan2_grid: ") + hipGetErrorString(e));
    return AFIS_OK;
}

int afis_debug_graph_arith(afis_ctx* ctx, unsigned long long* out8)
{
    if (!ctx || !out8) return fail(ctx, AFIS_EINVAL, "afis_debug_graph_arith: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    HIPCHK(ctx, d.ensure(64));
    hipError_t e = launch_debug_graph_arith(d.as<unsigned long long>(), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out8, d.p, 64, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d.release();
    if (e != hipSuccess) return fail(ctx, AFIS_EDEVICE, std::string("afis_debug_graph_arith: ") + hipGetErrorString(e));
    return AFIS_OK;
}

int afis_debug_lut(afis_ctx* ctx, const afis_template_view* query, float* out, int32_t* n_rows)
{
    if (!ctx || !query || !out) return fail(ctx, AFIS_EINVAL, "afis_debug_lut: null argument");
    if (query->n_tex <= 0) { if (n_rows) *n_rows = 0; return AFIS_OK; }
    const afis_texture_view& x = query->tex[0];
    if (x.des_len != kDes || !x.des) return fail(ctx, AFIS_EINVAL, "afis_debug_lut: latent texture template needs fp32 descriptors of length 96");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf des, lut;
    std::vector<float> h(x.des, x.des + (size_t)x.n * kDes);
    HIPCHK(ctx, upload(des, h, ctx->stream));
    HIPCHK(ctx, lut.ensure((size_t)x.n * kM * kK * 4));
    HIPCHK(ctx, launch_lut_reference_layout(des.as<float>(), x.n, ctx->codewords.as<float>(), lut.as<float>(), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, lut.p, (size_t)x.n * kM * kK * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    des.release(); lut.release();
    if (n_rows) *n_rows = x.n;
    return AFIS_OK;
}

int afis_debug_texture_rowmax(afis_ctx* ctx, const afis_template_view* query, int64_t gidx, float* val, int32_t* arg, int32_t* n_rows)
{
    if (!ctx || !query || !val || !arg) return fail(ctx, AFIS_EINVAL, "afis_debug_texture_rowmax: null argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_debug_texture_rowmax: commit the gallery first");
    if (gidx < 0 || gidx >= ctx->gal.G) return fail(ctx, AFIS_EINVAL, "afis_debug_texture_rowmax: gallery index out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    QueryGroup grp; std::vector<int32_t> st;
    int rc = build_group(ctx, query, 1, grp, st);
    if (rc != AFIS_OK) { grp.release(); return rc; }
    const QueryDev& d = grp.dev;
    const int n_lt = grp.h_lt_n[0];
    if (n_rows) *n_rows = n_lt;
    if (n_lt > 0) {
        const size_t n_pairs = (size_t)ctx->gal.G;
        HIPCHK(ctx, ctx->lut.ensure((size_t)d.n_tiles * kTileFloats * 4));
        HIPCHK(ctx, ctx->rm_val.ensure(n_pairs * d.lt_pad * 4));
        HIPCHK(ctx, ctx->rm_arg.ensure(n_pairs * d.lt_pad * 4));
        HIPCHK(ctx, hipMemsetAsync(ctx->rm_val.p, 0, n_pairs * d.lt_pad * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->rm_arg.p, 0, n_pairs * d.lt_pad * 4, ctx->stream));
        if (ctx->adc_variant == 9) { int rc9 = adc_stage_mfma(ctx, grp, true); if (rc9 != AFIS_OK) { grp.release(); return rc9; } }
        else if (ctx->adc_variant == 8) { int rc16 = adc_stage_q(ctx, grp, ctx->chunk > 0 ? ctx->chunk : 32, true); if (rc16 != AFIS_OK) { grp.release(); return rc16; } }
        else {
        { int rcf = ensure_codes_cf(ctx, ctx->adc_variant); if (rcf != AFIS_OK) { grp.release(); return rcf; } }
        HIPCHK(ctx, launch_lut_build(d, ctx->codewords.as<float>(), ctx->lut.as<float>(), ctx->adc_variant, ctx->stream));
        HIPCHK(ctx, launch_adc_rowmax(d, ctx->gal, ctx->lut.as<float>(), ctx->chunk > 0 ? ctx->chunk : 32, ctx->adc_variant, ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), ctx->stream));
        }
        HIPCHK(ctx, hipMemcpyAsync(val, ctx->rm_val.as<float>() + (size_t)gidx * d.lt_pad, (size_t)n_lt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(arg, ctx->rm_arg.as<int32_t>() + (size_t)gidx * d.lt_pad, (size_t)n_lt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    grp.release();
    return AFIS_OK;
}

// Parity tap: the correspondence list of one (latent, gallery template) pair after a stage of a scorer.
//   which 0 = texture scorer, 1..3 = minutiae scorer of selected template 27 / 3 / 12;  stage 0 = candidates (S3 / S7),
//   1 = after the distance filter (S8), 2 = after the angle filter (S9).  *n = -1 when the scorer is not run for the pair.
int afis_debug_stage_list(afis_ctx* ctx, const afis_template_view* query, int64_t gidx, int which, int stage,
                          float* sim, int32_t* li, int32_t* ri, int32_t* n)
{
    if (!ctx || !query || !sim || !li || !ri || !n || which < 0 || which > 3 || stage < 0 || stage > 2) return fail(ctx, AFIS_EINVAL, "afis_debug_stage_list: bad argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_debug_stage_list: commit the gallery first");
    if (gidx < 0 || gidx >= ctx->gal.G) return fail(ctx, AFIS_EINVAL, "afis_debug_stage_list: gallery index out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    QueryGroup grp; std::vector<int32_t> st;
    int rc = build_group(ctx, query, 1, grp, st);
    if (rc != AFIS_OK) { grp.release(); return rc; }
    *n = -1;
    DevBuf d_out, d_n;
    auto body = [&]() -> int {
        if (st[0] != AFIS_QUERY_OK) return AFIS_OK;
        const QueryDev& d = grp.dev;
        GalleryDev one = ctx->gal;
        one.G = 1; one.minu_off += gidx; one.minu_tile_off += gidx; one.tex_off += gidx; one.tex_cf_blk += gidx; one.empty += gidx;
        hipStream_t s = ctx->stream;
        HIPCHK(ctx, d_out.ensure(3 * (size_t)kTopTex * sizeof(MinuCand)));
        HIPCHK(ctx, d_n.ensure(3 * 4));
        HIPCHK(ctx, hipMemsetAsync(d_n.p, 0xff, 12, s));
        HIPCHK(ctx, ctx->parts.ensure(16));
        int slot = 0, cap = kTopTex;
        if (which == 0) {
            if (d.n_tiles <= 0) return AFIS_OK;
            HIPCHK(ctx, ctx->lut.ensure((size_t)d.n_tiles * kTileFloats * 4));
            HIPCHK(ctx, ctx->rm_val.ensure((size_t)d.lt_pad * 4)); HIPCHK(ctx, ctx->rm_arg.ensure((size_t)d.lt_pad * 4));
            const int av = ctx->adc_variant >= 8 ? 0 : ctx->adc_variant;     // the tap always uses a direct exact kernel (same bits); for the
            { int rcf = ensure_codes_cf(ctx, av); if (rcf != AFIS_OK) return rcf; }  // bound + refine variants the plain one, which needs no extra code stream
            one.tex_codes_cf = ctx->gal.tex_codes_cf;
            HIPCHK(ctx, launch_lut_build(d, ctx->codewords.as<float>(), ctx->lut.as<float>(), av, s));
            HIPCHK(ctx, launch_adc_rowmax(d, one, ctx->lut.as<float>(), 32, av, ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), s));
            HIPCHK(ctx, launch_graph_texture(d, one, ctx->table.as<float>(), ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), nullptr, nullptr, ctx->parts.as<float>(),
                                             d_out.as<MinuCand>(), d_n.as<int32_t>(), stage | (ctx->s89_tie_order << 8), s));
        } else {
            slot = which - 1; cap = kTopMinu;
            const size_t per_wg = minu_scratch_floats(grp.max_nL, ctx->max_nR, ctx->s3_tie_order);
            HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * 64));
            HIPCHK(ctx, ctx->cands.ensure(3 * (size_t)kTopMinu * sizeof(MinuCand))); HIPCHK(ctx, ctx->cand_n.ensure(12)); HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(3, 1) * 4));
            HIPCHK(ctx, launch_minu_cands(d, one, ctx->scratch.as<float>(), per_wg, 64, ctx->minu_generic | (ctx->s3_tie_order << 1), ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, nullptr, s));
            HIPCHK(ctx, launch_graph_minutiae(d, one, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->parts.as<float>(), nullptr, nullptr,
                                              d_out.as<MinuCand>(), d_n.as<int32_t>(), stage | (ctx->s89_tie_order << 8), s));
        }
        std::vector<MinuCand> h((size_t)3 * kTopTex); int32_t hn[3] = {-1, -1, -1};
        HIPCHK(ctx, hipMemcpyAsync(h.data(), d_out.p, h.size() * sizeof(MinuCand), hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(hn, d_n.p, 12, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        if (ctx->hg.empty[(size_t)gidx]) return AFIS_OK;                   // rolled empty: no scorer runs
        *n = hn[slot];
        for (int t = 0; t < hn[slot]; ++t) { const MinuCand& c = h[(size_t)slot * cap + t]; sim[t] = c.sim; li[t] = c.li; ri[t] = c.ri; }
        return AFIS_OK;
    };
    rc = body();
    d_out.release(); d_n.release(); grp.release();
    return rc;
}


}  // extern "C"
