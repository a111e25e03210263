// afis_search.cpp — the search side of the C ABI (include/afis_matcher.h): query groups resident on the device, the launch sequence of a search over the resident
// shard (the body of the reference's OpenMP gallery loop, matching/matcher.cpp:168-190 == :273-295, for a batch of latents), rank lists, correspondence export
// (matcher.cpp:321-327) and the all-templates mode (matcher.cpp:339-374).  Kernels: adc*.hip, minu.hip, graph.hip.
#include "afis_ctx.h"

using namespace afis;

namespace afis {

// Device bytes one latent of a launch group costs at worst (1000 texture rows): row maxima (value + point: 8 B per (pair, row) — adc_variant 9 keeps them in the compact list only, the others in the dense arrays),
// adc_variant 9's bound-pass records (kMfRecBytes per (template, row)), the minutiae candidate lists and the per-part scores.
int64_t group_bytes_per_query(const afis_ctx* ctx, int64_t G)
{
    const int64_t per_pair = (int64_t)kTexMax * 8 + (ctx->adc_variant == 9 ? (int64_t)kTexMax * kMfRecBytesPerRow : 0) + 3 * (int64_t)kTopMinu * (int64_t)sizeof(MinuCand) + 3 * 4 + 16 + 8;
    return std::max<int64_t>(1, G) * per_pair;
}
// ... and what a group of nq latents with rows_total latent texture rows, the longest of them lt_max, really takes (round 6: the buffers are sized by this, not by the worst case —
// the bench's latents have 671 rows on average, and the records of a 50-latent group at a 100k-template shard are 27 GB instead of 40)
int64_t group_bytes_actual(const afis_ctx* ctx, int64_t G, int64_t nq, int64_t rows_total, int64_t lt_max)
{
    const int64_t lt_pad = std::max<int64_t>(kTileRows, (lt_max + kTileRows - 1) / kTileRows * kTileRows), R_pad = (rows_total + 31) / 32 * 32;
    const int64_t per_pair = lt_pad * 8 + 3 * (int64_t)kTopMinu * (int64_t)sizeof(MinuCand) + 3 * 4 + 16 + 8;
    return std::max<int64_t>(1, G) * (nq * per_pair + (ctx->adc_variant == 9 ? R_pad * kMfRecBytesPerRow : 0));
}

// Contexts of this process, per device: what each of them holds in per-group buffers and what it has PLANNED to hold (its largest uploaded launch group) — a context that
// uploads its queries sees the free memory of the moment, and several contexts on one device each took 60 % of it before any of them had allocated a byte (round-4 advisor item).
namespace { std::mutex g_ctx_mutex; std::vector<afis_ctx*> g_ctx_live; }
static size_t held_group_bytes(const afis_ctx* c)
{
    return c->rm_val.bytes + c->rm_arg.bytes + c->rm_cv.bytes + c->rm_n.bytes + c->mf_rec.bytes + c->cands.bytes + c->cand_n.bytes + c->parts.bytes + c->minu_fb.bytes;
}
void register_context(afis_ctx* c) { std::lock_guard<std::mutex> lk(g_ctx_mutex); g_ctx_live.push_back(c); }
void unregister_context(afis_ctx* c) { std::lock_guard<std::mutex> lk(g_ctx_mutex); g_ctx_live.erase(std::remove(g_ctx_live.begin(), g_ctx_live.end(), c), g_ctx_live.end()); }

// What a launch group may take: the option, or 60 % of the free device memory (buffers this context already holds for earlier groups are reused, so they count as free;
// what OTHER contexts of the process on this device have planned but not yet allocated does not).
int64_t group_budget_bytes(const afis_ctx* ctx)
{
    if (ctx->rowmax_budget_bytes > 0) return ctx->rowmax_budget_bytes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 36ll << 30;
    int64_t pending_elsewhere = 0;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        for (const afis_ctx* c : g_ctx_live)
            if (c != ctx && c->device == ctx->device) pending_elsewhere += std::max<int64_t>(0, c->planned_group_bytes - (int64_t)held_group_bytes(c));
    }
    const int64_t avail = (int64_t)free_b + (int64_t)held_group_bytes(ctx) - pending_elsewhere;
    return std::max<int64_t>(1ll << 30, (int64_t)((double)std::max<int64_t>(0, avail) * 0.6));
}

// Host wait for streams with a deadline: hipStreamQuery on each of them in turn (which also keeps every one of them submitting: with ROCm 7.2 a blocking hipStreamSynchronize
// of the context's stream ALONE was seen not to return while work it depended on sat on the CU-masked side streams — round 4; round 5 saw it once more, in afis_destroy after an
// early return; tools/repro/README.md), a yield between rounds
// and a short sleep once the wait is long.  A device that does not come back within search_timeout_s is reported as AFIS_EDEVICE instead of holding the caller's
// thread for ever; when that happens with side streams in use, the context stops using them (bound_cus off: one stream, the kernels back to back).
int wait_streams(afis_ctx* ctx, std::initializer_list<hipStream_t> streams, const char* what)
{
    if (ctx->search_timeout_s <= 0) {                                      // unbounded: every stream of the list in turn (the side streams come first)
        static const bool last_only = AFIS_EXPERIMENT_ENV("AFIS_WAIT_CTX_SYNC_ONLY") != nullptr;     // experiment (tools/repro/README.md): round 4's hanging form, a blocking wait on the context's stream alone
        hipStream_t last = nullptr; for (hipStream_t st : streams) last = st;
        for (hipStream_t st : streams) if (st && (!last_only || st == last)) HIPCHK(ctx, hipStreamSynchronize(st));
        return AFIS_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; ++spins) {
        bool all = true;
        for (hipStream_t st : streams) {
            if (!st) continue;
            const hipError_t e = hipStreamQuery(st);
            if (e == hipErrorNotReady) all = false;
            else if (e != hipSuccess) return fail(ctx, AFIS_EDEVICE, std::string(what) + ": hipStreamQuery: " + hipGetErrorString(e));
        }
        if (all) return AFIS_OK;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ctx->search_timeout_s) {     // (looked at every round: hipStreamQuery itself returns in microseconds — tools/repro/side_stream_hang.hip, mode 5)
            if (streams.size() > 1) ctx->overlap_failed = true;
            ctx->search_abandoned = true;
            char msg[256];
            snprintf(msg, sizeof msg, "%s: the device did not finish within %.3g s (AFIS_SEARCH_TIMEOUT_S)%s", what, ctx->search_timeout_s,
                     streams.size() > 1 ? "; the overlapped schedule is switched off for this context (bound_cus 0)" : "");
            return fail(ctx, AFIS_EDEVICE, msg);
        }
        if (spins < 20000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// A search that left at its deadline may still be running on the device: what it queued must be done before its buffers (device, pinned) are touched again and before anything
// blocks on the context's stream alone.  Called at the top of the entry points that queue work.
int drain_abandoned(afis_ctx* ctx)
{
    if (!ctx->search_abandoned) return AFIS_OK;
    ctx->search_abandoned = false;
    const int rc = wait_streams(ctx, {ctx->stream_lo, ctx->stream_hi, ctx->stream}, "waiting for the search that timed out");   // (a second timeout sets the flag again)
    if (rc != AFIS_OK) return rc;
    // the device is idle: the query groups the abandoned search was still reading can go now (freeing them at its deadline would have been a hipFree under running
    // kernels — or an unbounded implicit device synchronisation, the very wait the deadline exists to avoid)
    for (afis_queries* q : ctx->parked_queries) { for (QueryGroup& g : q->groups) g.release(); delete q; }
    ctx->parked_queries.clear();
    return AFIS_OK;
}

// Work queued on the side streams must not outlive a failing search (it reads and writes the context's buffers): armed when the first kernel goes to a side stream,
// disarmed by the group's own wait; every early return in between drains both streams (bounded).
struct SideStreamGuard {
    afis_ctx* ctx; hipStream_t a = nullptr, b = nullptr; bool armed = false;
    explicit SideStreamGuard(afis_ctx* c) : ctx(c) {}
    void arm(hipStream_t x, hipStream_t y) { a = x; b = y; armed = true; }
    void disarm() { armed = false; }
    // (its own short bound — the launch error is what the caller must see, not a second full deadline — and the drain's outcome appended to the message)
    ~SideStreamGuard()
    {
        if (!armed) return;
        const std::string keep = ctx->err;
        const double full = ctx->search_timeout_s;
        if (full > 0) ctx->search_timeout_s = std::min(full, 10.0);
        const int rc = wait_streams(ctx, {a, b}, "draining the side streams after a failed launch group");
        ctx->search_timeout_s = full;
        ctx->err = rc == AFIS_OK ? keep : keep + " [and the side streams did not drain within 10 s: " + ctx->err + "]";
    }
};

}  // namespace afis

// ---------------------------------------------------------------------------------------------------------------------
namespace afis {

static const int kSelected[3] = {27 - 1, 3 - 1, 12 - 1};                   // matcher.cpp:380

// spec == NULL: the reference's selection for every query (templates 27, 3, 12 and texture template 0, matcher.cpp:380-415).
// spec != NULL (afis_match_all_templates): query i uses latent minutiae templates spec[i*4 + 0..2] (-1 = none) and latent texture
// template spec[i*4 + 3] (-1 = none), and is never "latent empty".
int build_group(afis_ctx* ctx, const afis_template_view* qs, int nq, QueryGroup& grp, std::vector<int32_t>& status_out, const int* spec)
{
    std::vector<int32_t> lm_off{0}, lt_off{0}, tile_off{0}, tile16_off{0}, tex_slot, status;
    std::vector<short2> lm_xy, lt_xy; std::vector<float> lm_ori, lm_des, lt_ori, lt_des;
    int max_nL = 0, lt_max = 0;
    for (int i = 0; i < nq; ++i) {
        const afis_template_view& t = qs[i];
        if (t.n_minu < 0 || t.n_tex < 0 || (t.n_minu > 0 && !t.minu) || (t.n_tex > 0 && !t.tex)) return fail(ctx, AFIS_EINVAL, "latent template: bad view");
        const int* sel = spec ? spec + (size_t)i * 4 : kSelected;
        const int tex_ind = spec ? spec[(size_t)i * 4 + 3] : 0;
        const bool latent_empty = !spec && (t.n_minu <= sel[0] && t.n_tex <= 0);     // matcher.cpp:383-386
        status.push_back(latent_empty ? AFIS_QUERY_LATENT_EMPTY : AFIS_QUERY_OK);
        for (int s = 0; s < 3; ++s) {
            if (!latent_empty && sel[s] >= 0 && t.n_minu > sel[s]) {
                const afis_minutiae_view& m = t.minu[sel[s]];
                if (m.n <= 0 || m.n > 2000 || !m.x || !m.y || !m.ori || !m.des) return fail(ctx, AFIS_EINVAL, "latent minutiae template: bad view (n must be 1..2000)");
                if (m.des_len != kDes) return fail(ctx, AFIS_EINVAL, "latent minutiae template: des_len must be 96 (the reference asserts equal descriptor lengths, matcher.cpp:433)");
                for (int k = 0; k < m.n; ++k) lm_xy.push_back(make_short2(m.x[k], m.y[k]));
                lm_ori.insert(lm_ori.end(), m.ori, m.ori + m.n);
                lm_des.insert(lm_des.end(), m.des, m.des + (size_t)m.n * kDes);
                max_nL = std::max(max_nL, m.n);
            }
            lm_off.push_back((int32_t)lm_xy.size());
        }
        int n_lt = 0;
        if (!latent_empty && tex_ind >= 0 && t.n_tex > tex_ind) {
            const afis_texture_view& x = t.tex[tex_ind];
            if (x.n <= 0 || x.n > 2000 || !x.x || !x.y || !x.ori || !x.des) return fail(ctx, AFIS_EINVAL, "latent texture template: bad view (n must be 1..2000, des required)");
            if (x.des_len != kDes) return fail(ctx, AFIS_EINVAL, "latent texture template: des_len must be 96");
            n_lt = std::min(x.n, kTexMax);                                   // matcher.cpp:544-545
            for (int k = 0; k < n_lt; ++k) lt_xy.push_back(make_short2(x.x[k], x.y[k]));
            lt_ori.insert(lt_ori.end(), x.ori, x.ori + n_lt);
            lt_des.insert(lt_des.end(), x.des, x.des + (size_t)n_lt * kDes);
        }
        lt_off.push_back((int32_t)lt_xy.size());
        tile_off.push_back(tile_off.back() + (n_lt + kTileRows - 1) / kTileRows);
        tile16_off.push_back(tile16_off.back() + (n_lt + 15) / 16);
        tex_slot.push_back(tex_ind >= 0 && t.n_tex > tex_ind ? t.n_minu : -1);
        lt_max = std::max(lt_max, n_lt);
        grp.h_lt_n.push_back(n_lt);
    }
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, upload(grp.lm_off, lm_off, s)); HIPCHK(ctx, upload(grp.lm_xy, lm_xy, s)); HIPCHK(ctx, upload(grp.lm_ori, lm_ori, s));
    HIPCHK(ctx, upload(grp.lm_des, lm_des, s)); HIPCHK(ctx, upload(grp.lt_off, lt_off, s));
    std::vector<int32_t> lm_tile_off;
    const std::vector<float> lm_frag = fragment_tiles(lm_des, lm_off, lm_tile_off);
    HIPCHK(ctx, upload(grp.lm_frag, lm_frag, s)); HIPCHK(ctx, upload(grp.lm_tile_off, lm_tile_off, s)); HIPCHK(ctx, upload(grp.lt_xy, lt_xy, s));
    HIPCHK(ctx, upload(grp.lt_ori, lt_ori, s)); HIPCHK(ctx, upload(grp.lt_des, lt_des, s)); HIPCHK(ctx, upload(grp.tile_off, tile_off, s)); HIPCHK(ctx, upload(grp.tile16_off, tile16_off, s));
    HIPCHK(ctx, upload(grp.tex_slot, tex_slot, s)); HIPCHK(ctx, upload(grp.status, status, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    QueryDev& d = grp.dev;
    d.nq = nq;
    d.lm_off = grp.lm_off.as<int32_t>(); d.lm_xy = grp.lm_xy.as<short2>(); d.lm_ori = grp.lm_ori.as<float>(); d.lm_des = grp.lm_des.as<float>(); d.lm_frag = grp.lm_frag.as<float4>(); d.lm_tile_off = grp.lm_tile_off.as<int32_t>();
    d.lt_off = grp.lt_off.as<int32_t>(); d.lt_xy = grp.lt_xy.as<short2>(); d.lt_ori = grp.lt_ori.as<float>(); d.lt_des = grp.lt_des.as<float>();
    d.tile_off = grp.tile_off.as<int32_t>(); d.tex_slot = grp.tex_slot.as<int32_t>(); d.status = grp.status.as<int32_t>();
    d.n_tiles = tile_off.back();
    d.tile16_off = grp.tile16_off.as<int32_t>(); d.n_tiles16 = tile16_off.back(); grp.n_lt_rows = lt_off.back();
    d.lt_pad = std::max(kTileRows, (lt_max + kTileRows - 1) / kTileRows * kTileRows);
    grp.nq = nq; grp.max_nL = max_nL; grp.n_lm_points = (int64_t)lm_xy.size();
    status_out.insert(status_out.end(), status.begin(), status.end());
    return AFIS_OK;
}

}  // namespace afis

extern "C" {

int afis_queries_upload(afis_ctx* ctx, const afis_template_view* queries, int n_q, afis_queries** out)
{
    if (!ctx || !out || n_q < 0 || (n_q > 0 && !queries)) return fail(ctx, AFIS_EINVAL, "afis_queries_upload: bad argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_queries_upload: commit the gallery first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { const int rcd = drain_abandoned(ctx); if (rcd != AFIS_OK) return rcd; }
    // group size: bounded by the option and by the memory budget of a group's per-pair buffers
    const int64_t G = std::max<int64_t>(1, ctx->gal.G);
    const int64_t by_mem = group_budget_bytes(ctx) / group_bytes_per_query(ctx, G);
    // latents per launch group: the option, or (0 = auto) as many as keep about five million (latent, rolled) pairs in a launch (round 5; two million before) — 50 at a 100k-template
    // shard, 128 at <= 39k (round 3, a 12.5k-template shard: 100 latents in one launch 310.7 ms, in 64 + 36: 315.1): the persistent per-pair kernels lose their tails once per launch, which shows on small shards (12 launches of 100k pairs
    // each cost 1.2 x their share of a 100k-template step; 2 launches do not).  Measured at 100k templates, 100 latents: 7 per launch 2 495 ms, 10: 2 486,
    // 15: 2 466, 20: 2 463, 34: 2 468.  Round 5, in the overlapped schedule (a launch group's bound pass beside its minutiae stage, the whole chip for what follows: every group ends in a
    // hand-over between the three streams): 12 per launch 2 059.8 ms, 17: 2 049.8, 20 (the dynamic programme cuts 100 latents into 6 x 16.7): 2 045.4, 25: 2 043.4, 34: 2 037.6, 50: 2 029.3 / 2 035.4,
    // 64 (64 + 36): 2 048.1, 100: 2 047.5 (profiles/r05_group_size_sweep.txt; two boxes, two passes each) — about five million pairs per launch now: 50 at a 100k-template shard.
    const int64_t want = ctx->query_batch > 0 ? ctx->query_batch : launch_group_latents(G);
    int per = (int)std::max<int64_t>(1, std::min<int64_t>(want, by_mem));
    afis_queries* q = new afis_queries();
    q->n_q = n_q;
    const int64_t budget = group_budget_bytes(ctx);
    // Launch groups are contiguous runs of at most `per` queries; with the matrix-core bound pass (adc_variant 9) the cuts are placed where its row groups of 768 latent
    // texture rows are fewest (launch_group_cuts, afis_device.h; tests/test_host.py checks the rule on the CPU).  Results do not depend on the cuts.
    std::vector<long long> rows((size_t)n_q + 1, 0);
    for (int i = 0; i < n_q; ++i) {
        const afis_template_view& t = queries[i];
        const bool has = t.n_tex > 0 && t.tex && !(t.n_minu <= kSelected[0] && t.n_tex <= 0);
        rows[(size_t)i + 1] = rows[(size_t)i] + (has ? std::min(std::max(t.tex[0].n, 0), kTexMax) : 0);
    }
    std::vector<int> cuts((size_t)std::max(n_q, 1));                        // group ends (exclusive)
    int n_cuts = 0;
    // `per` so far is what the budget allows if every latent had 1000 texture rows.  The latents are known here: when that worst case is what limits the group, larger groups are tried
    // against what they would REALLY take (group_bytes_actual), largest first.
    auto largest_group_bytes = [&](int n) {
        int64_t worst = 0; int a0 = 0;
        for (int c = 0; c < n; ++c) {
            int64_t lt_max = 0;
            for (int i = a0; i < cuts[(size_t)c]; ++i) lt_max = std::max<int64_t>(lt_max, rows[(size_t)i + 1] - rows[(size_t)i]);
            worst = std::max(worst, group_bytes_actual(ctx, G, cuts[(size_t)c] - a0, rows[(size_t)cuts[(size_t)c]] - rows[(size_t)a0], lt_max));
            a0 = cuts[(size_t)c];
        }
        return worst;
    };
    for (int try_per = (int)std::max<int64_t>(per, std::min<int64_t>(want, std::max(n_q, 1)));; --try_per) {
        launch_group_cuts(rows.data(), n_q, try_per, ctx->adc_variant == 9, cuts.data(), &n_cuts);
        if (try_per <= per || largest_group_bytes(n_cuts) <= budget) { per = try_per; break; }
    }
    ctx->planned_group_bytes = std::max(ctx->planned_group_bytes, largest_group_bytes(n_cuts));
    cuts.resize((size_t)n_cuts);
    int g0 = 0;
    for (int end : cuts) {
        q->groups.emplace_back();
        int rc = build_group(ctx, queries + g0, end - g0, q->groups.back(), q->status);
        if (rc != AFIS_OK) { afis_queries_free(ctx, q); return rc; }
        g0 = end;
    }
    *out = q;
    return AFIS_OK;
}

void afis_queries_free(afis_ctx* ctx, afis_queries* q)
{
    if (!q) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (ctx && ctx->search_abandoned) { ctx->parked_queries.push_back(q); return; }     // a search that left at its deadline may still read these buffers: freed once the device is back (drain_abandoned / afis_destroy)
    for (QueryGroup& g : q->groups) g.release();
    delete q;
}

// variants 6 / 7 read the gallery's codes from their own lane-ordered stream: lay it out now if this is their first use
}  // extern "C"

namespace afis {

#ifdef AFIS_EXPERIMENTAL_KERNELS
int ensure_codes_cf(afis_ctx* ctx, int variant)
{
    if ((variant != 6 && variant != 7) || ctx->codes_cf_built) return AFIS_OK;
    HIPCHK(ctx, ctx->g_tex_codes_cf.ensure(std::max<size_t>((size_t)ctx->cf_blocks * 64 * 16, 16)));
    ctx->gal.tex_codes_cf = ctx->g_tex_codes_cf.as<uint4>();
    HIPCHK(ctx, launch_codes_cf(ctx->gal, ctx->g_tex_codes_cf.p, ctx->stream));
    ctx->codes_cf_built = true;
    return AFIS_OK;
}
#endif

static int tile_share_of(const afis_ctx* ctx) { return ctx->tile_share > 0 ? ctx->tile_share : 4; }

// S4 + S5 + S6 of adc_variant 8 for one query group (rm_val / rm_arg sized by the caller): the quantised pass bounds the candidates, the fp32
// table (reference layout, all rows of the group) settles them
int adc_stage_q(afis_ctx* ctx, QueryGroup& grp, int chunk, bool exact, hipEvent_t after_lut)
{
    const QueryDev& d = grp.dev;
    hipStream_t s = ctx->stream;
    if (d.n_tiles16 <= 0 || ctx->gal.G <= 0) { if (after_lut) HIPCHK(ctx, hipEventRecord(after_lut, s)); return AFIS_OK; }
    if (!ctx->codes_q_built) {
        HIPCHK(ctx, ctx->g_tex_codes_q.ensure(std::max<size_t>((size_t)ctx->q_blocks * 64 * 16, 16)));
        HIPCHK(ctx, launch_codes_q(ctx->gal, ctx->g_tex_q_blk.as<int32_t>(), ctx->g_tex_codes_q.p, s));
        ctx->codes_q_built = true;
    }
    HIPCHK(ctx, ctx->lutq.ensure((size_t)d.n_tiles16 * 131072));
    HIPCHK(ctx, ctx->lutq_min.ensure(std::max<size_t>((size_t)grp.n_lt_rows * kM * 4, 16)));
    HIPCHK(ctx, ctx->lutq_rng.ensure(std::max<size_t>((size_t)grp.n_lt_rows * kM * 4, 16)));
    HIPCHK(ctx, ctx->lutq_rowc.ensure(std::max<size_t>((size_t)grp.n_lt_rows * 16, 16)));
    HIPCHK(ctx, launch_lutq_build(d, grp.n_lt_rows, ctx->codewords.as<float>(), ctx->lutq_min.as<float>(), ctx->lutq_rng.as<float>(), ctx->lutq.p, ctx->lutq_rowc.p, s));
    if (exact) {
        HIPCHK(ctx, ctx->lut32.ensure((size_t)grp.n_lt_rows * kM * kK * 4));
        HIPCHK(ctx, launch_lut_reference_layout(d.lt_des, grp.n_lt_rows, ctx->codewords.as<float>(), ctx->lut32.as<float>(), s));
    }
    if (after_lut) HIPCHK(ctx, hipEventRecord(after_lut, s));
    HIPCHK(ctx, launch_adc_rowmax_q(d, ctx->gal, ctx->g_tex_codes_q.p, ctx->g_tex_q_blk.as<int32_t>(), ctx->lutq.p, ctx->lutq_rowc.p,
                                    exact ? ctx->lut32.as<float>() : nullptr, chunk, tile_share_of(ctx), ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), s));
    return AFIS_OK;
}

// adc_variant 9's derived data: the codebook in fp16 with its squared norms (once per context) and the gallery's PQ codes as tiles of 32 points with their point terms
// (once per committed gallery).  Built by afis_gallery_commit when variant 9 is selected then — a resident gallery includes them — and on first use otherwise.
int ensure_mf_gallery(afis_ctx* ctx, hipStream_t s)
{
    const GalleryDev& g = ctx->gal;
    if (!ctx->mf_cb_built) {
        HIPCHK(ctx, ctx->mf_cw16.ensure((size_t)kM * kK * 16));
        HIPCHK(ctx, ctx->mf_cwn.ensure((size_t)kM * kK * 4));
        HIPCHK(ctx, launch_mf_codebook(ctx->codewords.as<float>(), ctx->mf_cw16.p, ctx->mf_cwn.as<float>(), s));
        ctx->mf_cb_built = true;
    }
    if (!ctx->mf_gal_built && g.G > 0) {
        const size_t n_ent = std::max<size_t>((size_t)ctx->t32_tiles * 32, 1);
        HIPCHK(ctx, ctx->g_codes_p.ensure(n_ent * 16));
        HIPCHK(ctx, ctx->g_nrm_p.ensure(n_ent * 4));
        HIPCHK(ctx, ctx->g_tile_meta.ensure(std::max<size_t>((size_t)ctx->t32_tiles * 8, 16)));
        HIPCHK(ctx, launch_mf_tiles(g, ctx->g_tex_t32_blk.as<int32_t>(), ctx->mf_cwn.as<float>(), ctx->g_codes_p.p, ctx->g_nrm_p.as<float>(), ctx->g_tile_meta.p, s));
        ctx->mf_gal_built = true;
    }
    return AFIS_OK;
}

// S4-S6 (+ the row selection of S7) of adc_variant 9 for one query group: row constants, matrix-core bound pass, selection by bounds and exact
// recomputation.  all_rows: every row is evaluated exactly (parity taps); otherwise rows that cannot reach the pair's top 200 get -inf.
// sb: the stream of the row constants and the bound pass (the context's stream, or the CU-masked one); refine_now false: the caller launches the selection / recomputation kernel itself (adc_refine_mfma)
int adc_stage_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, hipEvent_t after_lut, hipEvent_t after_bound, bool compact, hipStream_t sb, bool refine_now, unsigned long long* diag)
{
    const QueryDev& d = grp.dev;
    hipStream_t s = sb ? sb : ctx->stream;
    const GalleryDev& g = ctx->gal;
    if (grp.n_lt_rows <= 0 || g.G <= 0) { if (after_lut) HIPCHK(ctx, hipEventRecord(after_lut, s)); if (after_bound) HIPCHK(ctx, hipEventRecord(after_bound, s)); return AFIS_OK; }
    { int rcg = ensure_mf_gallery(ctx, s); if (rcg != AFIS_OK) return rcg; }
    const int n_rows = grp.n_lt_rows, n_rb = (n_rows + 31) / 32, R_pad = n_rb * 32;
    // The per-row buffers have been brought to the size of the search's LARGEST group by afis_search_resident before it queued anything: these calls find them large enough
    // (a hipMalloc behind queued work was seen to take 0.5-0.8 s; see there).  Callers outside a search (the parity taps) allocate here.
    const size_t R_cap = (size_t)R_pad;                                 // the group's own rows (afis_search_resident has brought the buffers to the largest group of the search)
    HIPCHK(ctx, ctx->mf_bfrag.ensure(R_cap / 32 * 6 * 64 * 16));
    HIPCHK(ctx, ctx->mf_rowk.ensure(R_cap * 16));
    HIPCHK(ctx, ctx->mf_rec.ensure((size_t)g.G * R_cap * kMfRecBytesPerRow));
    if (ctx->mf_collect_stats && !ctx->mf_stats.p) { HIPCHK(ctx, ctx->mf_stats.ensure(64)); HIPCHK(ctx, hipMemsetAsync(ctx->mf_stats.p, 0, 64, s)); }
    HIPCHK(ctx, launch_mf_rows(d.lt_des, n_rows, n_rb, ctx->codewords.as<float>(), ctx->mf_cwn.as<float>(), ctx->mf_bfrag.p, ctx->mf_rowk.p, s));
    if (after_lut) HIPCHK(ctx, hipEventRecord(after_lut, s));
    // workgroups = row groups x gallery chunks: about 24 per CU (a CU runs one at a time: the end of the launch idles at most ~1/24 of it),
    // a chunk never below 8 templates
    const int wg_rb = 24;                                              // row blocks per workgroup (adc_mfma.hip)
    const int n_rg = (n_rb + wg_rb - 1) / wg_rb;
    const long long want_chunks = (std::max<long long>(1, (256 * 24) / n_rg) + 7) / 8 * 8;   // a multiple of 8: the kernel gives every XCD its own chunks (adc_mfma.hip), an uneven count would leave XCDs idle at the end
    const int chunk = ctx->chunk > 0 ? ctx->chunk : (int)std::max<long long>(8, ((long long)g.G + want_chunks - 1) / want_chunks);
    HIPCHK(ctx, launch_adc_mfma(g, ctx->g_codes_p.p, ctx->g_nrm_p.as<float>(), ctx->g_tile_meta.p, ctx->g_tex_t32_blk.as<int32_t>(), ctx->mf_cw16.p,
                                ctx->mf_bfrag.p, ctx->mf_rowk.p, n_rows, n_rb, R_pad, chunk, ctx->mf_blocks, ctx->mf_rec.p, diag, s));
    if (after_bound) HIPCHK(ctx, hipEventRecord(after_bound, s));
    return refine_now ? adc_refine_mfma(ctx, grp, all_rows, compact) : AFIS_OK;
}

int adc_refine_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, bool compact)
{
    if (grp.n_lt_rows <= 0 || ctx->gal.G <= 0) return AFIS_OK;
    static const bool skip = AFIS_EXPERIMENT_ENV("AFIS_ABLATE_SKIP_TEXTURE_TAIL") != nullptr;    // timing experiments with ablated bound-pass builds (tools/r05_run10.sh): their records are garbage, the kernels behind the pass must not read them
    if (skip) return AFIS_OK;
    const int R_pad = (grp.n_lt_rows + 31) / 32 * 32;
    HIPCHK(ctx, launch_tex_refine(grp.dev, ctx->gal, ctx->codewords.as<float>(), ctx->mf_rec.p, ctx->mf_rowk.p, R_pad, all_rows ? 1 : 0, ctx->rm_val.as<float>(),
                                  ctx->rm_arg.as<int32_t>(), ctx->mf_collect_stats ? ctx->mf_stats.as<unsigned long long>() : nullptr,
                                  compact ? ctx->rm_cv.as<float>() : nullptr, compact ? ctx->rm_n.as<int32_t>() : nullptr, ctx->stream));
    return AFIS_OK;
}

// Rank lists are made on the device for k <= kDeviceTopK (k passes of a workgroup-wide maximum per query); larger k sorts on the host.
static const int kDeviceTopK = 64;

// (1 of 3) Every buffer of the launch groups, brought to its size while the device is idle and before anything of the search is queued.
static int prepare_search_buffers(afis_ctx* ctx, const afis_queries* q, bool want_parts)
{
    const int64_t G = ctx->gal.G;
    const int nq_all = q->n_q;
    float* const parts = want_parts ? reinterpret_cast<float*>(1) : nullptr;     // (only its being asked for matters here)
    // Every buffer of the launch groups is brought to its size HERE, while the device is idle and before anything of this search is queued: for the largest group of
    // the search (round 6: for what that group really takes — its own texture rows — not for the worst case of 1000 rows per latent), so that the calls further down never
    // re-allocate.  A hipMalloc of 6-13 GB takes 0.3 ms on an idle device; issued behind queued work (the row records used to be allocated inside adc_stage_mfma, after
    // the group's first kernels) it took 510-790 ms in three runs of ten (match -ldir: one search call in seven; profiles/r04_alloc_trace.txt).
    if (G > 0) {
        int nq_max = 0, nL_max = 1; size_t rows_pad_max = 32, pair_rows_max = 0;
        for (const QueryGroup& grp : q->groups) {
            nq_max = std::max(nq_max, grp.nq); nL_max = std::max(nL_max, grp.max_nL);
            rows_pad_max = std::max(rows_pad_max, ((size_t)std::max(grp.n_lt_rows, 0) + 31) / 32 * 32);
            pair_rows_max = std::max(pair_rows_max, (size_t)grp.nq * (size_t)grp.dev.lt_pad);          // a group's row-maximum arrays: [pair][lt_pad]
        }
        const size_t n_pairs = (size_t)nq_max * G;
        if (n_pairs > 0) {
            const size_t rm_bytes = std::max<size_t>(pair_rows_max * (size_t)G * 4, 16);
            if (ctx->adc_variant != 9) HIPCHK(ctx, ctx->rm_val.ensure(rm_bytes));      // variant 9: the values live in the compact list (rm_cv) only
            HIPCHK(ctx, ctx->rm_arg.ensure(rm_bytes));
            HIPCHK(ctx, ctx->parts.ensure(parts ? (size_t)nq_all * G * 16 : n_pairs * 16));   // per-part scores on request: every group's block stays on the device until the search is done
            HIPCHK(ctx, ctx->cands.ensure(n_pairs * 3 * kTopMinu * sizeof(MinuCand)));
            HIPCHK(ctx, ctx->cand_n.ensure(n_pairs * 3 * 4));
            HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(n_pairs * 3, (size_t)G) * 4));
            {   // the generic candidate kernel's scratch (sized as in the loop below, for the longest latent minutiae template of the search)
                const size_t per_wg = minu_scratch_floats(nL_max, ctx->max_nR, ctx->s3_tie_order);
                int n_wg = 1024;
                while (n_wg > 64 && per_wg * 4 * n_wg > (8ull << 30)) n_wg /= 2;
                HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * n_wg));
            }
            if (ctx->adc_variant == 9) {
                HIPCHK(ctx, ctx->rm_cv.ensure(rm_bytes)); HIPCHK(ctx, ctx->rm_n.ensure(n_pairs * 4));
                const size_t R_cap = rows_pad_max;
                HIPCHK(ctx, ctx->mf_bfrag.ensure(R_cap / 32 * 6 * 64 * 16));
                HIPCHK(ctx, ctx->mf_rowk.ensure(R_cap * 16));
                HIPCHK(ctx, ctx->mf_rec.ensure((size_t)G * R_cap * kMfRecBytesPerRow));
                if (!ctx->mf_gal_built) {                                  // first search: the bound pass's copy of the gallery codes (adc_stage_mfma fills it)
                    const size_t n_ent = std::max<size_t>((size_t)ctx->t32_tiles * 32, 1);
                    HIPCHK(ctx, ctx->g_codes_p.ensure(n_ent * 16));
                    HIPCHK(ctx, ctx->g_nrm_p.ensure(n_ent * 4));
                    HIPCHK(ctx, ctx->g_tile_meta.ensure(std::max<size_t>((size_t)ctx->t32_tiles * 8, 16)));
                }
            }
        }
    }
    return AFIS_OK;
}

// (3 of 3) What the search's events and diagnostics rows say: where the candidate tasks went, the clocks the sampled workgroups saw, the stage times per launch group.
static int collect_timing(afis_ctx* ctx, const afis_queries* q, afis_timing& tm, bool dev_topk, hipEvent_t* evk)
{
    const int64_t G = ctx->gal.G;
    const size_t n_groups = q->groups.size();
    {   // where the candidate tasks went, and the clocks the sampled workgroups saw (shader cycles per tick of the constant 100 MHz counter)
        unsigned long long acc[kDiagWords] = {};
        for (size_t i = 0; i < n_groups; ++i) for (int w = 0; w < kDiagWords; ++w) acc[w] += ctx->h_diag[i * kDiagWords + w];
        tm.minu_fallback_tasks = (int64_t)acc[kDiagFallback];
        tm.minu_tasks_small = (int64_t)acc[kDiagSmall]; tm.minu_tasks_medium = (int64_t)acc[kDiagSmall + 1]; tm.minu_tasks_large = (int64_t)acc[kDiagSmall + 2];
        tm.minu_tasks = tm.minu_tasks_small + tm.minu_tasks_medium + tm.minu_tasks_large + tm.minu_fallback_tasks;
        tm.cands_clock_ghz = acc[kDiagCandsWall] ? (float)((double)acc[kDiagCandsClk] / (double)acc[kDiagCandsWall] * 0.1) : 0.0f;
        tm.bound_clock_ghz = acc[kDiagBoundWall] ? (float)((double)acc[kDiagBoundClk] / (double)acc[kDiagBoundWall] * 0.1) : 0.0f;
    }
    if (G > 0) {
        for (size_t i = 0; i < n_groups; ++i) {
            hipEvent_t* ev = &ctx->evpool[i * 10];
            float tot = 0;
            auto el = [&](int a, int b, float& out) -> int { out = 0; HIPCHK(ctx, hipEventElapsedTime(&out, ev[a], ev[b])); return AFIS_OK; };
            const bool ov = q->groups[i].overlapped;
            float t_lut = 0, t_adc = 0, t_tex = 0, t_minu = 0, t_fuse = 0, t_bound = 0, t_ref = 0, t_c = 0, t_g = 0;
            if (el(0, 5, tot)) return AFIS_EDEVICE;
            if (ov) {                                                          // overlapped form: the bound pass's time is its own stream's, the minutiae stage ran beside it; the stage times overlap (their sum exceeds total_ms)
                if (el(0, 1, t_lut) || el(1, 6, t_bound) || el(8, 2, t_ref) || el(2, 3, t_tex) || el(0, 7, t_c) || el(7, 4, t_g) || el(9, 5, t_fuse)) return AFIS_EDEVICE;
                t_adc = t_bound + t_ref; t_minu = t_c + t_g;
            } else {
                if (el(0, 1, t_lut) || el(1, 2, t_adc) || el(2, 3, t_tex) || el(3, 4, t_minu) || el(9, 5, t_fuse) || el(3, 7, t_c) || el(7, 4, t_g)) return AFIS_EDEVICE;
                if (ctx->adc_variant == 9 && q->groups[i].n_lt_rows > 0) { if (el(1, 6, t_bound) || el(6, 2, t_ref)) return AFIS_EDEVICE; }
                else t_bound = t_adc;
            }
            tm.adc_bound_ms += t_bound; tm.adc_refine_ms += t_ref; tm.cands_ms += t_c; tm.minu_graph_ms += t_g;
            tm.lut_ms += t_lut; tm.adc_ms += t_adc; tm.tex_tail_ms += t_tex; tm.minu_ms += t_minu; tm.fuse_ms += t_fuse; tm.total_ms += tot;
        }
        if (dev_topk) { float t = 0; HIPCHK(ctx, hipEventElapsedTime(&t, evk[0], evk[1])); tm.topk_ms = t; tm.total_ms += t; }
    }
    return AFIS_OK;
}

}  // namespace afis

extern "C" {

int afis_search_resident(afis_ctx* ctx, afis_queries* q, float* scores, float* parts, int32_t* status,
                         int k, int64_t* topk_idx, float* topk_score)
{
    if (!ctx || !q) return fail(ctx, AFIS_EINVAL, "afis_search_resident: null argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_search: commit the gallery first");
    if (k < 0 || (k > 0 && (!topk_idx || !topk_score))) return fail(ctx, AFIS_EINVAL, "afis_search: k > 0 needs topk_idx and topk_score");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { const int rcd = drain_abandoned(ctx); if (rcd != AFIS_OK) return rcd; }
    const GalleryDev& g = ctx->gal;
    const int64_t G = g.G;
    const int nq_all = q->n_q;
    afis_timing tm = {};
    if (status) for (int i = 0; i < nq_all; ++i) status[i] = q->status[i];
    hipStream_t s = ctx->stream;
    // The groups run back to back on the stream(s): no host round trip between them.  Scores of ALL queries stay on the device
    // ([n_q][G]) for the rank-list kernel; they cross PCIe only when the caller asks for them.
    const size_t n_groups = q->groups.size();
    while (ctx->evpool.size() < n_groups * 10 + 2) { hipEvent_t e; HIPCHK(ctx, hipEventCreate(&e)); ctx->evpool.push_back(e); }
    if (G > 0 && nq_all > 0) HIPCHK(ctx, ctx->scores.ensure((size_t)nq_all * G * 4));
    HIPCHK(ctx, ctx->diag.ensure(std::max<size_t>(n_groups, 1) * kDiagWords * 8));
    HIPCHK(ctx, hipMemsetAsync(ctx->diag.p, 0, std::max<size_t>(n_groups, 1) * kDiagWords * 8, s));    // before the first group's ev[0]: ordered before everything the side streams do
    { const int rcp = prepare_search_buffers(ctx, q, parts != nullptr); if (rcp != AFIS_OK) return rcp; }
    static const bool alloc_trace = getenv("AFIS_ALLOC_TRACE") != nullptr;   // (the trace of DevBuf::ensure, afis_ctx.h: from this line on a search must not allocate — tests/test_gpu_parity.py)
    if (alloc_trace) { fprintf(stderr, "queue: the search starts queuing\n"); fflush(stderr); }
    int q0 = 0;
    size_t gi = 0;
    SideStreamGuard side_guard(ctx);
    bool any_overlap = false;
    static const bool skip_tex_tail = AFIS_EXPERIMENT_ENV("AFIS_ABLATE_SKIP_TEXTURE_TAIL") != nullptr;   // timing experiments only (see adc_refine_mfma)
    for (QueryGroup& grp : q->groups) {
        const QueryDev& d = grp.dev;
        const int nq = grp.nq;
        hipEvent_t* ev = &ctx->evpool[gi * 10];
        unsigned long long* const diag_row = ctx->diag.as<unsigned long long>() + gi * kDiagWords;
        if (G > 0) {
            const size_t n_pairs = (size_t)nq * G;
#ifdef AFIS_EXPERIMENTAL_KERNELS
            if (ctx->adc_variant < 8) HIPCHK(ctx, ctx->lut.ensure(std::max<size_t>((size_t)d.n_tiles * kTileFloats * 4, 16)));   // tile LUT of the direct kernels only
#endif
            const size_t lt_cap = (size_t)d.lt_pad;                          // (already large enough: the top of the search sized them for its largest group)
            if (ctx->adc_variant != 9) HIPCHK(ctx, ctx->rm_val.ensure(std::max<size_t>(n_pairs * lt_cap * 4, 16)));
            HIPCHK(ctx, ctx->rm_arg.ensure(std::max<size_t>(n_pairs * lt_cap * 4, 16)));
            if (ctx->adc_variant == 9) { HIPCHK(ctx, ctx->rm_cv.ensure(std::max<size_t>(n_pairs * lt_cap * 4, 16))); HIPCHK(ctx, ctx->rm_n.ensure(std::max<size_t>(n_pairs * 4, 16))); }
            HIPCHK(ctx, ctx->parts.ensure(parts ? (size_t)nq_all * G * 16 : n_pairs * 16));
            // minutiae scratch per workgroup: simi[n] | keys[n] | rowsum[2048] | colsum[2048]  (only pairs the fast kernel cannot take use it)
            size_t per_wg = minu_scratch_floats(grp.max_nL, ctx->max_nR, ctx->s3_tie_order);
            int n_wg = 1024;
            while (n_wg > 64 && per_wg * 4 * n_wg > (8ull << 30)) n_wg /= 2;
            HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * n_wg));
            HIPCHK(ctx, ctx->cands.ensure(n_pairs * 3 * kTopMinu * sizeof(MinuCand)));
            HIPCHK(ctx, ctx->cand_n.ensure(n_pairs * 3 * 4));
            HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(n_pairs * 3, (size_t)G) * 4));
            float* grp_scores = ctx->scores.as<float>() + (size_t)q0 * G;
            float* const grp_parts = ctx->parts.as<float>() + (parts ? (size_t)q0 * G * 4 : 0);
            // One ADC workgroup fills a CU (128 KB LUT tile), so nothing overlaps its tile load: chunks of ~640 templates keep that
            // under 3 % of a workgroup's life.  The blocks of XCD x are the chunks c % 8 == x, so the chunk COUNT is a multiple of 8
            // (measured at a 12.5k shard: 98 chunks of 128 -> 24 of 521: -9 % ADC time; at 100k: 196 of 512 -> 160 of 625: -2.5 %).
            // With tile_share s the blocks that follow one another on an XCD take s consecutive chunks against the SAME tile (8 instead of 32
            // tiles' fp32 tables — the refine's gathers — compete for an XCD's L2 at s = 4), so the count is a multiple of 8 s: -4.5 % ADC time
            // at 100k, -3 % at 12.5k.  (Round-2's first measurement of tile_share, with 196 chunks of 512, had shown a loss: the unbalanced
            // chunk count hid the gain.)
            const long long cmul = 8ll * (ctx->adc_variant == 8 ? tile_share_of(ctx) : 1);
            const long long n_chunks_auto = ((G + 639) / 640 + cmul - 1) / cmul * cmul;
            const int chunk = ctx->chunk > 0 ? ctx->chunk : (int)((G + n_chunks_auto - 1) / n_chunks_auto);
            HIPCHK(ctx, hipEventRecord(ev[0], s));
            // (a launch of fewer than 2^16 pairs — a single latent against 10k templates — is tail-bound, not power-bound: the side streams only add their hand-overs: 4.40 vs 4.54 ms)
            // ... and a group whose minutiae stage is much heavier than its bound pass (rolled prints of 130 +- 40 minutiae against latents of up to 150: bench.py --workload wide) loses:
            // the candidate kernels would stay confined to half of the chip long after the pass has ended (measured: 4 215 ms per step overlapped against 3 864 back to back).
            // The stage's work is priced by its similarity cells (latent x rolled minutiae) against the pass's (latent rows x rolled points): at the headline shapes the candidate
            // kernel alone takes 0.49 of the bound pass alone for 0.0179 of its cells; on half the CUs it takes twice that, so it still ends with the pass at about twice the headline's ratio.
            const double cells_m = (double)grp.n_lm_points * (double)ctx->total_minutiae, cells_t = (double)grp.n_lt_rows * (double)ctx->total_tex_points;
            const bool minutiae_light = cells_m <= ctx->overlap_cell_ratio * cells_t;
            const bool overlap = ctx->adc_variant == 9 && ctx->stream_lo != nullptr && !ctx->overlap_failed && n_pairs >= 65536 && minutiae_light;
            grp.overlapped = overlap;
            const bool compact9 = ctx->adc_variant == 9;                  // the recomputation kernel's compact list of the rows that matter (S7 reads a third of the rows)
            auto minutiae_stage = [&]() -> int {
                HIPCHK(ctx, launch_minu_cands(d, g, ctx->scratch.as<float>(), per_wg, n_wg, ctx->minu_generic | (ctx->s3_tie_order << 1), ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, diag_row, s));
                HIPCHK(ctx, hipEventRecord(ev[7], s));
                HIPCHK(ctx, launch_graph_minutiae(d, g, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), grp_parts, nullptr, nullptr, nullptr, nullptr, 2 | (ctx->s89_tie_order << 8), s));
                return AFIS_OK;
            };
            if (overlap) {
                // The bound pass is power-limited: half of the chip's CUs deliver 0.64 of the whole chip's matrix throughput (profiles/r04_cu_mask_probe.json).  It runs on a
                // stream confined to the low `bound_cus` CUs; the minutiae stage — candidates, then lists: independent of the texture path — runs beside it on a stream
                // confined to the OTHER CUs (an unconfined stream's persistent workgroups would take every CU and the bound pass, whose workgroup needs a whole CU's LDS,
                // would wait for them to leave).  When the bound pass is done the context's stream joins the list kernel (a second instance drawing from the same counter),
                // then runs recomputation and texture lists on the whole chip.
                hipStream_t sl = ctx->stream_lo, sh = ctx->stream_hi;
                side_guard.arm(sl, sh);
                HIPCHK(ctx, hipStreamWaitEvent(sl, ev[0], 0));                             // everything of the previous group (this stream's order) is done
                HIPCHK(ctx, hipStreamWaitEvent(sh, ev[0], 0));
                int rc9 = adc_stage_mfma(ctx, grp, false, ev[1], ev[6], true, sl, false, diag_row);
                if (rc9 != AFIS_OK) return rc9;
                HIPCHK(ctx, launch_minu_cands(d, g, ctx->scratch.as<float>(), per_wg, n_wg, ctx->minu_generic | (ctx->s3_tie_order << 1), ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, diag_row, sh));
                HIPCHK(ctx, hipMemsetAsync(g.task_ctr + 1, 0, 4, sh));                     // the list counter both instances of the list kernel draw from: reset BEFORE either may start
                HIPCHK(ctx, hipEventRecord(ev[7], sh));
                HIPCHK(ctx, launch_graph_minutiae(d, g, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), grp_parts, nullptr, nullptr, nullptr, nullptr, 2 | (ctx->s89_tie_order << 8), sh, true));
                HIPCHK(ctx, hipEventRecord(ev[4], sh));
                HIPCHK(ctx, hipStreamWaitEvent(s, ev[6], 0));
                HIPCHK(ctx, hipEventRecord(ev[8], s));                                     // the bound pass is done
                rc9 = adc_refine_mfma(ctx, grp, false, true);
                if (rc9 != AFIS_OK) return rc9;
                HIPCHK(ctx, hipEventRecord(ev[2], s));
                if (!skip_tex_tail) HIPCHK(ctx, launch_graph_texture(d, g, ctx->table.as<float>(), ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), ctx->rm_cv.as<float>(), ctx->rm_n.as<int32_t>(), grp_parts, nullptr, nullptr, 2 | (ctx->s89_tie_order << 8), s));
                HIPCHK(ctx, hipEventRecord(ev[3], s));
                HIPCHK(ctx, hipStreamWaitEvent(s, ev[7], 0));                              // every candidate list exists: help with whatever lists are left
                HIPCHK(ctx, launch_graph_minutiae(d, g, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), grp_parts, nullptr, nullptr, nullptr, nullptr, 2 | (ctx->s89_tie_order << 8), s, true));
                HIPCHK(ctx, hipStreamWaitEvent(s, ev[4], 0));
                // No host wait here: the groups of a search follow one another on the three streams through events alone, and the search's final wait polls ALL THREE streams
                // (wait_streams).  Round 4 blocked on the two side streams after every group because hipStreamSynchronize of the context's stream alone never returned with
                // ROCm 7.2 while work it depends on sat on the CU-masked side streams; a hipStreamQuery loop does return (profiles/r05_side_stream_waits.json: 46.11 / 46.09 / 46.03
                // queries/s without the group wait polling one stream / all three / with the group wait), and it is bounded.  AFIS_GROUP_WAIT=1 restores the per-group wait.
                static const bool group_wait = AFIS_EXPERIMENT_ENV("AFIS_GROUP_WAIT") != nullptr;
                if (group_wait) { const int rcw = wait_streams(ctx, {sl, sh}, "afis_search: side streams of a launch group"); side_guard.disarm(); if (rcw != AFIS_OK) return rcw; }
                any_overlap = true;
            } else {
            if (ctx->adc_variant == 9) {                                    // fp16 matrix-core bound pass + exact recomputation
                int rc9 = adc_stage_mfma(ctx, grp, false, ev[1], ev[6], true, nullptr, true, diag_row);
                if (rc9 != AFIS_OK) return rc9;
            } else if (ctx->adc_variant == 8) {                             // 16-bit fixed-point LDS-table bound pass + exact refine
                int rc16 = adc_stage_q(ctx, grp, chunk, true, ev[1]);
                if (rc16 != AFIS_OK) return rc16;
            } else {
#ifdef AFIS_EXPERIMENTAL_KERNELS                                              // the direct exact kernels (adc_direct.hip): test library only
                { int rcf = ensure_codes_cf(ctx, ctx->adc_variant); if (rcf != AFIS_OK) return rcf; }
                HIPCHK(ctx, launch_lut_build(d, ctx->codewords.as<float>(), ctx->lut.as<float>(), ctx->adc_variant, s));
                HIPCHK(ctx, hipEventRecord(ev[1], s));
                HIPCHK(ctx, launch_adc_rowmax(d, g, ctx->lut.as<float>(), chunk, ctx->adc_variant, ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), s));
#else
                return fail(ctx, AFIS_EINVAL, "adc_variant: the direct kernels are not part of this library");
#endif
            }
            HIPCHK(ctx, hipEventRecord(ev[2], s));
            if (!skip_tex_tail) HIPCHK(ctx, launch_graph_texture(d, g, ctx->table.as<float>(), ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), compact9 ? ctx->rm_cv.as<float>() : nullptr,
                                             compact9 ? ctx->rm_n.as<int32_t>() : nullptr, grp_parts, nullptr, nullptr, 2 | (ctx->s89_tie_order << 8), s));
            HIPCHK(ctx, hipEventRecord(ev[3], s));
            { int rcm = minutiae_stage(); if (rcm != AFIS_OK) return rcm; }
            HIPCHK(ctx, hipEventRecord(ev[4], s));
            }
            HIPCHK(ctx, hipEventRecord(ev[9], s));
            HIPCHK(ctx, launch_fuse(d, g, grp_parts, grp_scores, s));
            HIPCHK(ctx, hipEventRecord(ev[5], s));
            if (d.n_tiles > 0) {
                tm.adc_launches += 1;
                const int tile_rows = ctx->adc_variant == 8 ? 16 : kTileRows;   // rows the launched kernel pads a latent to (variant 9 does no table look-ups: the count is nominal there)
                int64_t rows = 0; for (int n : grp.h_lt_n) rows += (n + tile_rows - 1) / tile_rows * tile_rows;
                tm.adc_lookups += rows * ctx->total_tex_points * kM;
            }
            tm.pairs += (int64_t)n_pairs;
        }
        q0 += nq; ++gi;
    }
    // ---- rank lists (matcher.cpp:306-309; ties by ascending index) ----
    const bool dev_topk = k > 0 && k <= kDeviceTopK && G > 0 && nq_all > 0;
    hipEvent_t* evk = &ctx->evpool[n_groups * 10];
    if (dev_topk) {
        HIPCHK(ctx, ctx->topk_idx.ensure((size_t)nq_all * k * 8));
        HIPCHK(ctx, ctx->topk_score.ensure((size_t)nq_all * k * 4));
        HIPCHK(ctx, hipEventRecord(evk[0], s));
        HIPCHK(ctx, launch_topk(ctx->scores.as<float>(), nq_all, (int)G, k, (long long)ctx->index_base, ctx->topk_idx.as<long long>(), ctx->topk_score.as<float>(), s));
        HIPCHK(ctx, hipEventRecord(evk[1], s));
    }
    // What comes back inside the wait goes through a PINNED buffer of the context (rank lists, diagnostics: a few KB): an "asynchronous" copy into pageable host memory — the caller's
    // arrays, a std::vector — makes the runtime wait for the stream inside the call, which is where a search used to spend its two seconds before the deadline below was ever looked at
    // (tools/repro/timeout_recovery.py).  The caller's arrays are filled from it after the wait; the score matrix (-ldir: 40 MB at 100 x 100k) is copied after the wait as well.
    const size_t pin_topk = dev_topk ? (size_t)nq_all * k * 12 : 0, pin_diag = std::max<size_t>(n_groups, 1) * kDiagWords * 8;
    if (ctx->h_pin_bytes < pin_topk + pin_diag) {
        if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
        ctx->h_pin = nullptr; ctx->h_pin_bytes = 0;
        HIPCHK(ctx, hipHostMalloc(&ctx->h_pin, pin_topk + pin_diag, hipHostMallocDefault));
        ctx->h_pin_bytes = pin_topk + pin_diag;
    }
    uint8_t* const pin = (uint8_t*)ctx->h_pin;
    if (dev_topk) {
        HIPCHK(ctx, hipMemcpyAsync(pin, ctx->topk_idx.p, (size_t)nq_all * k * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(pin + (size_t)nq_all * k * 8, ctx->topk_score.p, (size_t)nq_all * k * 4, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipMemcpyAsync(pin + pin_topk, ctx->diag.p, pin_diag, hipMemcpyDeviceToHost, s));
    {
        const int rcw = any_overlap ? wait_streams(ctx, {ctx->stream_lo, ctx->stream_hi, s}, "afis_search") : wait_streams(ctx, {s}, "afis_search");
        side_guard.disarm();
        if (rcw != AFIS_OK) return rcw;
    }
    if (dev_topk) { memcpy(topk_idx, pin, (size_t)nq_all * k * 8); memcpy(topk_score, pin + (size_t)nq_all * k * 8, (size_t)nq_all * k * 4); }
    ctx->h_diag.assign(std::max<size_t>(n_groups, 1) * kDiagWords, 0ull);
    memcpy(ctx->h_diag.data(), pin + pin_topk, pin_diag);
    if (parts && G > 0 && nq_all > 0) HIPCHK(ctx, hipMemcpy(parts, ctx->parts.p, (size_t)nq_all * G * 16, hipMemcpyDeviceToHost));   // per-part scores on request (tests, the all-templates mode)
    const bool host_topk = k > 0 && !dev_topk;
    float* h_sc = scores;
    if (G > 0 && nq_all > 0 && (scores || host_topk)) {                    // the device is idle now: a plain copy
        if (!h_sc) { ctx->h_scores.resize((size_t)nq_all * G); h_sc = ctx->h_scores.data(); }
        HIPCHK(ctx, hipMemcpy(h_sc, ctx->scores.p, (size_t)nq_all * G * 4, hipMemcpyDeviceToHost));
    }
    { const int rct = collect_timing(ctx, q, tm, dev_topk, evk); if (rct != AFIS_OK) return rct; }
    if (host_topk) {                                                       // k > kDeviceTopK (or an empty gallery)
        std::vector<int32_t> ind((size_t)G);
        for (int i = 0; i < nq_all; ++i) {
            const float* sc = G > 0 ? h_sc + (size_t)i * G : nullptr;
            std::iota(ind.begin(), ind.end(), 0);
            const int kk = (int)std::min<int64_t>(k, G);
            std::partial_sort(ind.begin(), ind.begin() + kk, ind.end(), [sc](int a, int b) { return sc[a] > sc[b] || (sc[a] == sc[b] && a < b); });
            for (int r = 0; r < k; ++r) {
                const size_t o = (size_t)i * k + r;
                if (r < kk) { topk_idx[o] = ctx->index_base + ind[r]; topk_score[o] = sc[ind[r]]; }
                else { topk_idx[o] = -1; topk_score[o] = -INFINITY; }
            }
        }
    }
    tm.launch_groups = (int32_t)n_groups;
    for (const QueryGroup& grp : q->groups) tm.overlapped_groups += grp.overlapped ? 1 : 0;
    ctx->timing = tm;
    return AFIS_OK;
}

// Correspondence export (matcher.cpp:321-327 calling :376-417 with save_corr = true, :497-505): the minutiae scorers of the
// three selected latent templates are re-run against each listed gallery template with the kernels' survivor lists switched on.
int afis_correspondences(afis_ctx* ctx, const afis_template_view* query, const int64_t* gallery_idx, int n, int32_t* counts, int16_t* xy)
{
    if (!ctx || !query || n < 0 || (n > 0 && (!gallery_idx || !counts || !xy))) return fail(ctx, AFIS_EINVAL, "afis_correspondences: bad argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_correspondences: commit the gallery first");
    if (n == 0) return AFIS_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { const int rcd = drain_abandoned(ctx); if (rcd != AFIS_OK) return rcd; }
    const GalleryDev& g = ctx->gal;
    for (int i = 0; i < n; ++i)
        if (gallery_idx[i] < ctx->index_base || gallery_idx[i] >= ctx->index_base + g.G) return fail(ctx, AFIS_EINVAL, "afis_correspondences: gallery index outside this shard");
    QueryGroup grp;
    std::vector<int32_t> status;
    int rc = build_group(ctx, query, 1, grp, status);
    if (rc != AFIS_OK) { grp.release(); return rc; }
    for (int i = 0; i < n * 3; ++i) counts[i] = -1;
    memset(xy, 0, (size_t)n * 3 * kTopMinu * 4 * sizeof(int16_t));
    DevBuf d_xy, d_n;
    auto body = [&]() -> int {
        if (status[0] != AFIS_QUERY_OK) return AFIS_OK;                    // matcher.cpp:383-386: nothing is matched, nothing written
        const size_t per_wg = minu_scratch_floats(grp.max_nL, ctx->max_nR, ctx->s3_tie_order);
        const int n_wg = 64;
        HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * n_wg));
        HIPCHK(ctx, ctx->cands.ensure((size_t)n * 3 * kTopMinu * sizeof(MinuCand)));
        HIPCHK(ctx, ctx->cand_n.ensure((size_t)n * 3 * 4));
        HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(3, 1) * 4));
        HIPCHK(ctx, ctx->parts.ensure((size_t)n * 16));
        HIPCHK(ctx, d_xy.ensure((size_t)n * 3 * kTopMinu * sizeof(short4)));
        HIPCHK(ctx, d_n.ensure((size_t)n * 3 * 4));
        hipStream_t s = ctx->stream;
        int err = AFIS_OK;
        for (int i = 0; i < n && err == AFIS_OK; ++i) {
            const int64_t gi = gallery_idx[i] - ctx->index_base;
            GalleryDev one = g;                                            // a one-template view: offsets are absolute, so only the CSR bases move
            one.G = 1; one.minu_off += gi; one.minu_tile_off += gi; one.tex_off += gi; one.tex_cf_blk += gi; one.empty += gi;
            MinuCand* cands = ctx->cands.as<MinuCand>() + (size_t)i * 3 * kTopMinu;
            int32_t* cand_n = ctx->cand_n.as<int32_t>() + (size_t)i * 3;
            if (launch_minu_cands(grp.dev, one, ctx->scratch.as<float>(), per_wg, n_wg, ctx->minu_generic | (ctx->s3_tie_order << 1), cands, cand_n, ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, nullptr, s) != hipSuccess ||
                launch_graph_minutiae(grp.dev, one, cands, cand_n, ctx->parts.as<float>() + (size_t)i * 4,
                                      d_xy.as<short4>() + (size_t)i * 3 * kTopMinu, d_n.as<int32_t>() + (size_t)i * 3, nullptr, nullptr, 2 | (ctx->s89_tie_order << 8), s) != hipSuccess)
                err = fail(ctx, AFIS_EDEVICE, "afis_correspondences: kernel launch failed");
        }
        if (err == AFIS_OK) {
            if (hipMemcpyAsync(xy, d_xy.p, (size_t)n * 3 * kTopMinu * sizeof(short4), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(counts, d_n.p, (size_t)n * 3 * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                err = fail(ctx, AFIS_EDEVICE, "afis_correspondences: copy back failed");
            // -1 where the reference does not run the scorer at all (no file): rolled empty (:388-391), rolled without a
            // minutiae template (:399), latent without the selected template (:402-403)
            for (int i = 0; i < n && err == AFIS_OK; ++i) {
                const int64_t gi = gallery_idx[i] - ctx->index_base;
                int32_t off[2] = {0, 0};
                if (hipMemcpy(off, g.minu_off + gi, sizeof(off), hipMemcpyDeviceToHost) != hipSuccess) { err = fail(ctx, AFIS_EDEVICE, "afis_correspondences: copy back failed"); break; }
                for (int sl = 0; sl < 3; ++sl)
                    if (ctx->hg.empty[(size_t)gi] || off[1] - off[0] <= 0 || query->n_minu <= kSelected[sl]) counts[i * 3 + sl] = -1;
            }
        } else (void)hipStreamSynchronize(s);
        return err;
    };
    rc = body();
    d_xy.release(); d_n.release();
    grp.release();
    return rc;
}

// One2One_matching_all_templates (matcher.cpp:339-374) for one latent against the whole resident gallery: EVERY latent minutiae
// template vs rolled minutiae template 0 and EVERY latent texture template vs rolled texture template 0.  The kernels are the
// same; the latent is presented as ceil(max(n_minu/3, n_tex)) pseudo-queries whose three "selected" slots are templates
// 3j, 3j+1, 3j+2 and whose texture template is j, and the per-part scores are scattered back into the reference's score vector.
int afis_match_all_templates(afis_ctx* ctx, const afis_template_view* query, float* scores, int32_t* rolled_status, int32_t* query_status)
{
    if (!ctx || !query || !scores) return fail(ctx, AFIS_EINVAL, "afis_match_all_templates: null argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_match_all_templates: commit the gallery first");
    const int n_minu = query->n_minu, n_tex = query->n_tex;
    if (n_minu < 0 || n_tex < 0) return fail(ctx, AFIS_EINVAL, "afis_match_all_templates: bad view");
    const int64_t G = ctx->gal.G;
    const int width = n_minu + n_tex;
    if (query_status) *query_status = (n_minu <= 0 && n_tex <= 0) ? AFIS_QUERY_LATENT_EMPTY : AFIS_QUERY_OK;     // :345-348
    if (rolled_status) for (int64_t g = 0; g < G; ++g) rolled_status[g] = ctx->hg.empty[(size_t)g] ? 2 : 0;        // :350-353
    for (size_t i = 0; i < (size_t)G * width; ++i) scores[i] = 0.0f;                                             // :342-343
    if (width == 0 || G == 0) return AFIS_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { const int rcd = drain_abandoned(ctx); if (rcd != AFIS_OK) return rcd; }
    const int n_pq = std::max((n_minu + 2) / 3, n_tex);
    const int64_t by_mem = std::max<int64_t>(1, group_budget_bytes(ctx) / group_bytes_per_query(ctx, G));
    const int per = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->query_batch > 0 ? ctx->query_batch : 10, by_mem));
    std::vector<float> parts;
    for (int j0 = 0; j0 < n_pq; j0 += per) {
        const int nq = std::min(per, n_pq - j0);
        std::vector<afis_template_view> views((size_t)nq, *query);
        std::vector<int> spec((size_t)nq * 4);
        for (int j = 0; j < nq; ++j) {
            for (int s = 0; s < 3; ++s) spec[(size_t)j * 4 + s] = 3 * (j0 + j) + s < n_minu ? 3 * (j0 + j) + s : -1;
            spec[(size_t)j * 4 + 3] = j0 + j < n_tex ? j0 + j : -1;
        }
        afis_queries q; q.n_q = nq;
        q.groups.emplace_back();
        int rc = build_group(ctx, views.data(), nq, q.groups.back(), q.status, spec.data());
        if (rc == AFIS_OK) {
            parts.resize((size_t)nq * G * 4);
            rc = afis_search_resident(ctx, &q, nullptr, parts.data(), nullptr, 0, nullptr, nullptr);
        }
        if (ctx->search_abandoned) {                                        // timed out: the kernels may still read the group — park it (see afis_queries_free)
            afis_queries* keep = new afis_queries; keep->groups.swap(q.groups); ctx->parked_queries.push_back(keep);
        } else q.groups.back().release();
        if (rc != AFIS_OK) return rc;
        for (int j = 0; j < nq; ++j)
            for (int64_t g = 0; g < G; ++g) {
                if (ctx->hg.empty[(size_t)g]) continue;                      // rolled empty: the vector stays zero (return 2 before any scorer)
                const float* p = &parts[((size_t)j * G + g) * 4];
                float* o = scores + (size_t)g * width;
                for (int s = 0; s < 3; ++s) if (3 * (j0 + j) + s < n_minu) o[3 * (j0 + j) + s] = p[s];
                if (j0 + j < n_tex) o[n_minu + j0 + j] = p[3];
            }
    }
    return AFIS_OK;
}

int afis_search(afis_ctx* ctx, const afis_template_view* queries, int n_q, float* scores, float* parts, int32_t* status,
                int k, int64_t* topk_idx, float* topk_score)
{
    afis_queries* q = nullptr;
    int rc = afis_queries_upload(ctx, queries, n_q, &q);
    if (rc != AFIS_OK) return rc;
    rc = afis_search_resident(ctx, q, scores, parts, status, k, topk_idx, topk_score);
    afis_queries_free(ctx, q);
    return rc;
}

int afis_search_dat(afis_ctx* ctx, const void* const* latent_bytes, const size_t* lens, int n_q, float* scores, float* parts,
                    int32_t* status, int k, int64_t* topk_idx, float* topk_score)
{
    if (!ctx || n_q < 0 || (n_q > 0 && (!latent_bytes || !lens))) return fail(ctx, AFIS_EINVAL, "afis_search_dat: bad argument");
    std::vector<HostTemplate> ts(n_q);
    std::vector<std::vector<afis_minutiae_view>> mv(n_q);
    std::vector<std::vector<afis_texture_view>> tv(n_q);
    std::vector<afis_template_view> views(n_q);
    for (int i = 0; i < n_q; ++i) {
        (void)parse_latent_dat(latent_bytes[i], lens[i], ts[i]);           // the reference ignores this return code (matcher.cpp:150)
        views_of(ts[i], mv[i], tv[i], views[i]);
    }
    return afis_search(ctx, views.data(), n_q, scores, parts, status, k, topk_idx, topk_score);
}

}  // extern "C"
