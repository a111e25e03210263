// afis_api.cpp — implementation of the C ABI in include/afis_matcher.h: gallery packing (SoA), upload, query
// grouping and the launch sequence of the HIP kernels.  Host C++ only; device code lives in adc.hip, minu.hip, graph.hip and pq_encode.hip.
#include "../../include/afis_matcher.h"
#ifdef AFIS_PARITY_TAPS
#include "../../include/afis_matcher_taps.h"
#endif

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include <sys/mman.h>

#include "afis_device.h"
#include "template_io.h"

using namespace afis;

namespace {

thread_local std::string g_create_error;   // last afis_create failure of THIS thread (there is no context to hang it on)

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        static const bool trace = getenv("AFIS_ALLOC_TRACE") != nullptr;   // development aid: every (re)allocation of 64 MB or more, with the time it took, on stderr
        const auto t0 = std::chrono::steady_clock::now();
        const size_t was = bytes;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        const auto t1 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        if (trace && n >= ((size_t)64 << 20))
            fprintf(stderr, "alloc: %.3f GB (was %.3f): hipFree %.1f ms, hipMalloc %.1f ms\n", n / 1e9, was / 1e9, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};


}  // namespace

// One group of latents resident on the device.
struct QueryGroup {
    QueryDev dev;
    DevBuf lm_off, lm_xy, lm_ori, lm_des, lm_frag, lm_tile_off, lt_off, lt_xy, lt_ori, lt_des, tile_off, tile16_off, tex_slot, status;
    int nq = 0; int max_nL = 0; int n_lt_rows = 0; int64_t lut_rows_x_tiles = 0;
    int64_t n_lm_points = 0;             // latent minutiae of the group's three selected templates per query, summed
    bool overlapped = false;             // how the last search scheduled this group (afis_search_resident)
    std::vector<int32_t> h_lt_n;
    void release() { lm_off.release(); lm_xy.release(); lm_ori.release(); lm_des.release(); lm_frag.release(); lm_tile_off.release(); lt_off.release(); lt_xy.release(); lt_ori.release();
                     lt_des.release(); tile_off.release(); tile16_off.release(); tex_slot.release(); status.release(); }
};

struct afis_queries {
    std::vector<QueryGroup> groups;
    std::vector<int32_t> status;     // per query
    int n_q = 0;
};

struct afis_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_hi = nullptr;     // option bound_cus: the complement of stream_lo's CUs, for the minutiae stage while the bound pass runs
    hipStream_t stream_lo = nullptr;     // option bound_cus: a stream confined to the low N CUs (N / 8 of every XCD) for the power-limited bound pass; the rest of a launch group runs beside it
    int bound_cus = 0;                   // 0 = off: one stream, the kernels of a group back to back
    int n_cus = 0;                       // compute units of the device (hipDeviceProp_t::multiProcessorCount): the CU masks are built for this many
    std::vector<hipEvent_t> evpool;      // 10 per query group + 2: the groups of a search run back to back, timings are read at the end
    std::string err;
    DevBuf codewords, table;
    HostGallery hg;
    // afis_gallery_load into an empty staging area keeps the container MAPPED instead of copying its 50 KB per template into hg: the commit uploads the shard
    // [pend_first, pend_first + pend_count) straight from the mapping.  Anything else that touches the staged gallery first copies it into hg (materialise()).
    std::unique_ptr<GalleryMapping> pend;
    int64_t pend_first = 0, pend_count = 0;
    std::thread staging_reaper;          // returns the staged arrays to the system after the commit (0.5 s per 5 GB), off the caller's path; joined in afis_destroy
    bool committed = false;
    int64_t index_base = 0;
    GalleryDev gal;
    DevBuf g_minu_off, g_minu_xy, g_minu_ori, g_minu_des, g_minu_frag, g_minu_tile_off, g_tex_off, g_tex_xy, g_tex_ori, g_tex_codes, g_tex_codes_cf, g_tex_cf_blk, g_tex_codes_q, g_tex_q_blk, g_tex_t32_blk, g_empty, g_task_ctr;
    bool codes_cf_built = false;         // variants 6 / 7: their lane-ordered code stream, laid out on first use
    int64_t cf_blocks = 0;
    bool codes_q_built = false;          // adc_variant 8's lane-ordered code stream is laid out on first use
    int64_t q_blocks = 0;
    int64_t t32_tiles = 0;               // tiles of 32 rolled texture points (ceil(n/32) per template): the matrix-core bound pass's stream
    int max_nR = 0;
    int64_t total_tex_points = 0;
    int64_t total_minutiae = 0;          // rolled minutiae of the shard
    // adc_variant 9: fp16 codebook + |cw|^2 (once), pair-aligned gallery codes / point terms / pair directory (first use), per group B fragments,
    // row constants and the bound pass's records
    DevBuf mf_cw16, mf_cwn, g_codes_p, g_nrm_p, g_tile_meta, mf_bfrag, mf_rowk, mf_rec, mf_stats;
    bool mf_cb_built = false, mf_gal_built = false;
    int mf_collect_stats = 0;
    int mf_blocks = 2;                   // row blocks per wave of the bound pass: 2 (12 waves per workgroup) or 3 (8 waves, a third less LDS traffic per MFMA)
    DevBuf lutq, lutq_min, lutq_rng, lutq_rowc, lut32;      // adc_variant 8: 16-row fixed-point tiles, per-(row, m) min / range, per-row (offset, step, margin), fp32 table
    DevBuf lut, rm_val, rm_arg, rm_cv, rm_n, parts, scores, scratch, cands, cand_n, minu_fb, topk_idx, topk_score;
    DevBuf diag;                         // kDiagWords unsigned 64-bit counters per launch group of a search (afis_device.h): zeroed when the search starts, read back with its results
    std::vector<unsigned long long> h_diag;
    std::vector<float> h_scores, h_parts;
    int adc_variant = 9;                 // 9: fp16 matrix-core bound pass + exact recomputation (default); 8: 16-bit LDS-table bound pass + exact refine; 7: direct exact kernel; 0-3, 6: earlier direct kernels
    int tile_share = 0;                  // adc_variant 8: consecutive chunks per tile on an XCD; 0 = 4 (the refine's fp32 table stays in L2)
    int query_batch = 0;                 // latents per launch group at most; 0 = by shard size (afis_queries_upload); adc_variant 9 places the cuts by latent texture rows
    int chunk = 0;                       // gallery templates per ADC workgroup; 0 = by gallery size
    int minu_generic = 0;
    double search_timeout_s = 600.0;     // bound on every host wait of a search (AFIS_SEARCH_TIMEOUT_S; <= 0: plain hipStreamSynchronize, unbounded)
    bool overlap_failed = false;         // a wait of the overlapped schedule timed out: later searches keep to one stream
    double overlap_cell_ratio = 0.037;   // a launch group runs the overlapped schedule while (latent x rolled minutiae cells) <= this x (latent texture rows x rolled texture points); AFIS_OVERLAP_CELL_RATIO
    int64_t rowmax_budget_bytes = 0;     // device memory a launch group's per-pair buffers may take (option rowmax_budget_mb); 0 = 60 % of what hipMemGetInfo reports free
    afis_timing timing = {};
};

namespace {

int fail(afis_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

// the staged gallery as host arrays: a container that afis_gallery_load only mapped is copied into ctx->hg now
int materialise(afis_ctx* ctx)
{
    if (!ctx->pend) return AFIS_OK;
    std::string err;
    HostGallery add;
    if (!read_gallery_container(ctx->pend->path, ctx->pend_first, ctx->pend_count, add, nullptr, nullptr, err)) return fail(ctx, AFIS_EFORMAT, "gallery container: " + err);
    ctx->hg = std::move(add);
    ctx->pend.reset(); ctx->pend_first = ctx->pend_count = 0;
    return AFIS_OK;
}
#define HIPCHK(ctx, call)                                                                                       \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                          \
        return fail(ctx, AFIS_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// Host wait for streams with a deadline: hipStreamQuery on each of them in turn (which also keeps every one of them submitting: with ROCm 7.2 a hipStreamSynchronize
// of the context's stream ALONE never returned while work it depends on sat on the CU-masked side streams — tools/repro/side_stream_hang.hip), a yield between rounds
// and a short sleep once the wait is long.  A device that does not come back within search_timeout_s is reported as AFIS_EDEVICE instead of holding the caller's
// thread for ever; when that happens with side streams in use, the context stops using them (bound_cus off: one stream, the kernels back to back).
int wait_streams(afis_ctx* ctx, std::initializer_list<hipStream_t> streams, const char* what)
{
    if (ctx->search_timeout_s <= 0) {
        for (hipStream_t st : streams) if (st) HIPCHK(ctx, hipStreamSynchronize(st));
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
        if ((spins & 255) == 255 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ctx->search_timeout_s) {
            if (streams.size() > 1) ctx->overlap_failed = true;
            char msg[256];
            snprintf(msg, sizeof msg, "%s: the device did not finish within %.0f s (AFIS_SEARCH_TIMEOUT_S)%s", what, ctx->search_timeout_s,
                     streams.size() > 1 ? "; the overlapped schedule is switched off for this context (bound_cus 0)" : "");
            return fail(ctx, AFIS_EDEVICE, msg);
        }
        if (spins < 20000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
// Work queued on the side streams must not outlive a failing search (it reads and writes the context's buffers): armed when the first kernel goes to a side stream,
// disarmed by the group's own wait; every early return in between drains both streams (bounded).
struct SideStreamGuard {
    afis_ctx* ctx; hipStream_t a = nullptr, b = nullptr; bool armed = false;
    explicit SideStreamGuard(afis_ctx* c) : ctx(c) {}
    void arm(hipStream_t x, hipStream_t y) { a = x; b = y; armed = true; }
    void disarm() { armed = false; }
    ~SideStreamGuard() { if (armed) { const std::string keep = ctx->err; (void)wait_streams(ctx, {a, b}, "draining the side streams after a failed launch group"); ctx->err = keep; } }
};

template <class T, class A>
hipError_t upload(DevBuf& b, const std::vector<T, A>& v, hipStream_t s)
{
    hipError_t e = b.ensure(std::max<size_t>(v.size() * sizeof(T), 16));
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
}

// t->codes == NULL: `encoded` holds the PQ codes the device made from t->des (afis_gallery_add)
void append_entry(HostGallery& hg, const afis_minutiae_view* m, const afis_texture_view* t, const uint8_t* encoded = nullptr)
{
    if (m && m->n > 0) {
        hg.mx.insert(hg.mx.end(), m->x, m->x + m->n); hg.my.insert(hg.my.end(), m->y, m->y + m->n);
        hg.mori.insert(hg.mori.end(), m->ori, m->ori + m->n);
        hg.mdes.insert(hg.mdes.end(), m->des, m->des + (size_t)m->n * kDes);
    }
    hg.minu_off.push_back((int64_t)hg.mx.size());
    if (t && t->n > 0) {
        const int n = std::min(t->n, kTexMax);                              // matcher.cpp:546-547
        hg.tx.insert(hg.tx.end(), t->x, t->x + n); hg.ty.insert(hg.ty.end(), t->y, t->y + n);
        hg.tori.insert(hg.tori.end(), t->ori, t->ori + n);
        const uint8_t* codes = t->codes ? t->codes : encoded;
        hg.tcodes.insert(hg.tcodes.end(), codes, codes + (size_t)n * kM);
    }
    hg.tex_off.push_back((int64_t)hg.tx.size());
    hg.empty.push_back((!(m && m->n > 0) && !(t && t->n > 0)) ? 1 : 0);
}

int check_rolled(afis_ctx* ctx, const afis_template_view& t)
{
    if (t.n_minu < 0 || t.n_tex < 0 || (t.n_minu > 0 && !t.minu) || (t.n_tex > 0 && !t.tex)) return fail(ctx, AFIS_EINVAL, "rolled template: bad view");
    if (t.n_minu > 0) {
        const afis_minutiae_view& m = t.minu[0];
        if (m.n <= 0 || m.n > 2000 || !m.x || !m.y || !m.ori || !m.des) return fail(ctx, AFIS_EINVAL, "rolled minutiae template: bad view (n must be 1..2000)");
        if (m.des_len != kDes) return fail(ctx, AFIS_EINVAL, "rolled minutiae template: des_len must be 96");
    }
    if (t.n_tex > 0) {
        const afis_texture_view& x = t.tex[0];
        if (x.n <= 0 || x.n > 2000 || !x.x || !x.y || !x.ori || (!x.codes && !x.des)) return fail(ctx, AFIS_EINVAL, "rolled texture template: bad view (n must be 1..2000, codes or des required)");
        if (x.codes ? x.des_len != kM : x.des_len != kDes)
            return fail(ctx, AFIS_EINVAL, "rolled texture template: des_len must be 16 with PQ codes, 96 with fp32 descriptors (encoded on the device)");
    }
    return AFIS_OK;
}

void views_of(const HostTemplate& t, std::vector<afis_minutiae_view>& mv, std::vector<afis_texture_view>& tv, afis_template_view& out)
{
    mv.clear(); tv.clear();
    for (const HostMinutiae& m : t.minu) mv.push_back({m.n(), m.x.data(), m.y.data(), m.ori.data(), m.des_len, m.des.data()});
    for (const HostTexture& x : t.tex) tv.push_back({x.n(), x.x.data(), x.y.data(), x.ori.data(), x.des_len, x.des.empty() ? nullptr : x.des.data(), x.codes.empty() ? nullptr : x.codes.data()});
    out.n_minu = (int)mv.size(); out.minu = mv.data(); out.n_tex = (int)tv.size(); out.tex = tv.data();
}

// host-side re-layouts at commit touch every byte of the shard once: split [0, n) over a few threads
template <class F>
void parallel_for(int64_t n, F body)
{
    const int64_t nt = std::min<int64_t>(std::max<int64_t>(1, (int64_t)std::thread::hardware_concurrency()), std::min<int64_t>(16, std::max<int64_t>(1, n / 256)));
    if (nt <= 1) { body((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int64_t i = 0; i < nt; ++i) th.emplace_back(body, n * i / nt, n * (i + 1) / nt);
    for (std::thread& t : th) t.join();
}

// Descriptors re-laid as operand fragments of v_mfma_f32_16x16x4_f32 (minu.hip): template t (rows off[t] .. off[t+1]) becomes
// ceil(n/16) tiles of 6 x 64 float4; lane l of load v holds des[16*tile + (l&15)][4*(4v + c) + (l>>4)], c = 0..3.  Rows past the
// template's end are zero.  tile_off[t] = first tile of template t.
template <class Off>
std::vector<float> fragment_tiles(const std::vector<float>& des, const std::vector<Off>& off, std::vector<int32_t>& tile_off)
{
    const int64_t T = (int64_t)off.size() - 1;
    tile_off.assign((size_t)T + 1, 0);
    for (int64_t t = 0; t < T; ++t) tile_off[(size_t)t + 1] = tile_off[(size_t)t] + (int32_t)((off[(size_t)t + 1] - off[(size_t)t] + 15) / 16);
    std::vector<float> out((size_t)tile_off[(size_t)T] * 6 * 64 * 4, 0.0f);
    parallel_for(T, [&](int64_t lo, int64_t hi) {
        for (int64_t t = lo; t < hi; ++t) {
            const int64_t r0 = (int64_t)off[(size_t)t], n = (int64_t)off[(size_t)t + 1] - r0;
            for (int64_t row = 0; row < n; ++row) {
                const float* src = &des[(size_t)(r0 + row) * kDes];
                float* tile = &out[(size_t)(tile_off[(size_t)t] + row / 16) * 6 * 64 * 4];
                const int li = (int)(row & 15);
                for (int v = 0; v < 6; ++v)
                    for (int lg = 0; lg < 4; ++lg)
                        for (int c = 0; c < 4; ++c) tile[((size_t)v * 64 + lg * 16 + li) * 4 + c] = src[4 * (4 * v + c) + lg];
            }
        }
    });
    return out;
}

// Device bytes one latent of a launch group costs at worst (1000 texture rows): row maxima (value, point, compact list: 12 B per (pair, row)),
// adc_variant 9's bound-pass records (kMfRecBytes per (template, row)), the minutiae candidate lists and the per-part scores.
constexpr int64_t kMfRecBytesPerRow = 8;
int64_t group_bytes_per_query(const afis_ctx* ctx, int64_t G)
{
    const int64_t per_pair = (int64_t)kTexMax * 12 + (ctx->adc_variant == 9 ? (int64_t)kTexMax * kMfRecBytesPerRow : 0) + 3 * (int64_t)kTopMinu * (int64_t)sizeof(MinuCand) + 3 * 4 + 16 + 8;
    return std::max<int64_t>(1, G) * per_pair;
}
// What a launch group may take: the option, or 60 % of the free device memory (buffers this context already holds for earlier groups are reused, so they count as free).
int64_t group_budget_bytes(const afis_ctx* ctx)
{
    if (ctx->rowmax_budget_bytes > 0) return ctx->rowmax_budget_bytes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 36ll << 30;
    const size_t held = ctx->rm_val.bytes + ctx->rm_arg.bytes + ctx->rm_cv.bytes + ctx->rm_n.bytes + ctx->mf_rec.bytes + ctx->cands.bytes + ctx->cand_n.bytes + ctx->parts.bytes + ctx->minu_fb.bytes;
    return std::max<int64_t>(1ll << 30, (int64_t)((double)(free_b + held) * 0.6));
}

void free_gallery_dev(afis_ctx* c)
{
    c->g_minu_off.release(); c->g_minu_xy.release(); c->g_minu_ori.release(); c->g_minu_des.release(); c->g_minu_frag.release(); c->g_minu_tile_off.release();
    c->g_tex_off.release(); c->g_tex_xy.release(); c->g_tex_ori.release(); c->g_tex_codes.release(); c->g_tex_codes_cf.release(); c->g_tex_cf_blk.release(); c->g_tex_codes_q.release(); c->g_tex_q_blk.release(); c->g_tex_t32_blk.release(); c->g_empty.release(); c->g_task_ctr.release();
    c->g_codes_p.release(); c->g_nrm_p.release(); c->g_tile_meta.release(); c->mf_gal_built = false;
}

}  // namespace

extern "C" {

int afis_create(afis_ctx** out, const float* codewords, int M, int K, int dsub, int device_id)
{
    if (!out || !codewords) return fail(nullptr, AFIS_EINVAL, "afis_create: null argument");
    *out = nullptr;
    if (M != kM || K != kK || dsub != kDsub) return fail(nullptr, AFIS_EINVAL, "afis_create: only the M=16, K=256, dsub=6 codebook geometry is supported");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(nullptr, AFIS_EDEVICE, "afis_create: no HIP device (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n_dev) return fail(nullptr, AFIS_EINVAL, "afis_create: device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, AFIS_EDEVICE, "afis_create: hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, AFIS_EDEVICE, std::string("afis_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    afis_ctx* c = new afis_ctx();
    c->device = device_id;
    c->n_cus = prop.multiProcessorCount;
#define CRCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(e_); afis_destroy(c); return AFIS_EDEVICE; } } while (0)
    CRCHK(hipSetDevice(device_id));
    if (const char* cm = getenv("AFIS_CU_MASK")) {                           // experiment knob: comma-separated hex words of a CU mask for the context's stream
        std::vector<uint32_t> words;
        for (const char* p = cm; *p;) { words.push_back((uint32_t)strtoul(p, nullptr, 16)); const char* q = strchr(p, ','); if (!q) break; p = q + 1; }
        CRCHK(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)words.size(), words.data()));
    } else
    CRCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    std::vector<float> cw(codewords, codewords + (size_t)M * K * dsub);
    CRCHK(upload(c->codewords, cw, c->stream));
    std::vector<float> table((size_t)kDistN * kDistN);                     // matcher.cpp:45-56
    for (int i = 0; i < kDistN; ++i)
        for (int j = i; j < kDistN; ++j) {
            table[i * kDistN + j] = (float)sqrt((i * 16.0) * (i * 16.0) + (j * 16.0) * (j * 16.0));
            table[j * kDistN + i] = table[i * kDistN + j];
        }
    CRCHK(upload(c->table, table, c->stream));
    CRCHK(hipStreamSynchronize(c->stream));
#undef CRCHK
    *out = c;
    // The default schedule: the power-limited bound pass on half of the chip's CUs, the minutiae stage beside it on the other half (afis_search_resident; -7.6 % per step at 100k
    // templates, profiles/r04_overlap_ab.json).  AFIS_BOUND_CUS overrides (0 = one stream, the kernels back to back).  A runtime that refuses CU masks leaves it off.
    {
        if (const char* r = getenv("AFIS_OVERLAP_CELL_RATIO")) c->overlap_cell_ratio = atof(r);
        if (const char* r = getenv("AFIS_SEARCH_TIMEOUT_S")) c->search_timeout_s = atof(r);
        const char* e = getenv("AFIS_BOUND_CUS");
        const int64_t n = e ? atoll(e) : (c->n_cus == 256 ? 128 : 0);          // measured on the whole MI355X (256 CUs); a partitioned device keeps the single stream unless told otherwise
        if (afis_set_option(c, "bound_cus", n) != AFIS_OK) { c->bound_cus = 0; c->err.clear(); }
    }
    return AFIS_OK;
}

int afis_device_info(int device_id, char* name, size_t name_cap, char* pci_bus_id, size_t pci_cap, char* uuid_hex, size_t uuid_cap, int* n_cus)
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev) return AFIS_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return AFIS_EDEVICE;
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (pci_bus_id && pci_cap) { if (hipDeviceGetPCIBusId(pci_bus_id, (int)pci_cap, device_id) != hipSuccess) pci_bus_id[0] = 0; }
    if (uuid_hex && uuid_cap) {
        uuid_hex[0] = 0;
        hipUUID u;
        if (uuid_cap >= 33 && hipDeviceGetUuid(&u, device_id) == hipSuccess)
            for (int i = 0; i < 16; ++i) snprintf(uuid_hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)u.bytes[i]);
    }
    if (n_cus) *n_cus = prop.multiProcessorCount;
    return AFIS_OK;
}

int afis_create_from_codebook(afis_ctx** out, const void* bytes, size_t len, int device_id)
{
    HostCodebook cb;
    if (!bytes || !parse_codebook(bytes, len, cb)) return fail(nullptr, AFIS_EFORMAT, "codebook is empty!");
    return afis_create(out, cb.words.data(), cb.M, cb.K, cb.dsub, device_id);
}

void afis_destroy(afis_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipStream_t* ps : {&c->stream_lo, &c->stream_hi}) if (*ps) { (void)hipStreamSynchronize(*ps); (void)hipStreamDestroy(*ps); *ps = nullptr; }
    free_gallery_dev(c);
    c->codewords.release(); c->table.release(); c->lut.release(); c->rm_val.release(); c->rm_arg.release(); c->rm_cv.release(); c->rm_n.release();
    c->parts.release(); c->scores.release(); c->scratch.release(); c->cands.release(); c->cand_n.release(); c->minu_fb.release(); c->diag.release(); c->topk_idx.release(); c->topk_score.release(); c->lutq.release(); c->lutq_min.release(); c->lutq_rng.release(); c->lutq_rowc.release(); c->lut32.release();
    c->mf_cw16.release(); c->mf_cwn.release(); c->mf_bfrag.release(); c->mf_rowk.release(); c->mf_rec.release(); c->mf_stats.release();
    for (auto& e : c->evpool) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->staging_reaper.joinable()) c->staging_reaper.join();
    delete c;
}

const char* afis_last_error(const afis_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int afis_gallery_add(afis_ctx* ctx, const afis_template_view* t, int n)
{
    if (!ctx || (n > 0 && !t)) return fail(ctx, AFIS_EINVAL, "afis_gallery_add: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    for (int i = 0; i < n; ++i) { int rc = check_rolled(ctx, t[i]); if (rc) return rc; }
    std::vector<uint8_t> enc;
    for (int i = 0; i < n; ++i) {
        const afis_texture_view* x = t[i].n_tex > 0 ? &t[i].tex[0] : nullptr;
        if (x && !x->codes) {                                               // fp32 descriptors: PQ-encode on the device
            enc.resize((size_t)x->n * kM);
            int rc = afis_pq_encode(ctx, x->des, x->n, enc.data());
            if (rc != AFIS_OK) return rc;
        }
        append_entry(ctx->hg, t[i].n_minu > 0 ? &t[i].minu[0] : nullptr, x, enc.data());
    }
    return AFIS_OK;
}

// PQ encoder: TrainedPQEncoder.encode_multi (extraction/descriptor_PQ.py:19-27) on the device, in slices that fit a fixed
// staging buffer.
int afis_pq_encode(afis_ctx* ctx, const float* des, int64_t n, uint8_t* codes)
{
    if (!ctx || n < 0 || (n > 0 && (!des || !codes))) return fail(ctx, AFIS_EINVAL, "afis_pq_encode: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t slice = 1 << 20;                                          // 1 Mi points = 384 MiB of descriptors per launch
    DevBuf d_des, d_codes;
    int rc = AFIS_OK;
    for (int64_t i0 = 0; i0 < n && rc == AFIS_OK; i0 += slice) {
        const int64_t m = std::min(slice, n - i0);
        if (d_des.ensure((size_t)m * kDes * 4) != hipSuccess || d_codes.ensure((size_t)m * kM) != hipSuccess ||
            hipMemcpyAsync(d_des.p, des + i0 * kDes, (size_t)m * kDes * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            launch_pq_encode(d_des.as<float>(), m, ctx->codewords.as<float>(), d_codes.as<uint8_t>(), ctx->stream) != hipSuccess ||
            hipMemcpyAsync(codes + i0 * kM, d_codes.p, (size_t)m * kM, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(ctx, AFIS_EDEVICE, std::string("afis_pq_encode: ") + hipGetErrorString(hipGetLastError()));
    }
    d_des.release(); d_codes.release();
    return rc;
}

// The rolled branch of descriptor_PQ.py::encode_PQ (:332-349): a template whose texture descriptors are fp32 (the latent
// on-disk layout, descriptor_PQ.py:80-175) is rewritten in the rolled layout (:178-272) with every texture template's
// descriptors replaced by their PQ codes.
int afis_encode_rolled_dat(afis_ctx* ctx, const void* bytes, size_t len, void* out, size_t out_cap, size_t* out_len, int* load_rc)
{
    if (!ctx || !out_len || (len > 0 && !bytes)) return fail(ctx, AFIS_EINVAL, "afis_encode_rolled_dat: bad argument");
    HostTemplate t;
    const int rc = parse_latent_dat(bytes, len, t);
    if (load_rc) *load_rc = rc;
    if (rc < 0) { t.minu.clear(); t.tex.clear(); }
    for (HostTexture& x : t.tex) {
        if (x.des_len != kDes) return fail(ctx, AFIS_EINVAL, "afis_encode_rolled_dat: texture descriptors must be 96-d fp32");
        x.codes.resize((size_t)x.n() * kM);
        const int e = afis_pq_encode(ctx, x.des.data(), x.n(), x.codes.data());
        if (e != AFIS_OK) return e;
        x.des.clear(); x.des_len = kM;
    }
    const std::vector<uint8_t> w = write_rolled_dat(t);
    *out_len = w.size();
    if (!out || out_cap < w.size()) return out ? fail(ctx, AFIS_EINVAL, "afis_encode_rolled_dat: output buffer too small") : AFIS_OK;
    memcpy(out, w.data(), w.size());
    return AFIS_OK;
}

int afis_gallery_add_dat(afis_ctx* ctx, const void* bytes, size_t len, int* load_rc)
{
    if (!ctx) return AFIS_EINVAL;
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add_dat: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    HostTemplate t;
    int rc = parse_rolled_dat(bytes, len, t);
    // matcher.cpp:173-177: a negative code discards the template.  Code 8 (a descriptor length outside 1..192, where the reference overruns a
    // stack buffer) is this parser's own: the cursor is misaligned from there on, so the partial template is discarded too (score -1).
    if (rc < 0 || rc == 8) { t.minu.clear(); t.tex.clear(); }
    if (load_rc) *load_rc = rc;
    std::vector<afis_minutiae_view> mv; std::vector<afis_texture_view> tv; afis_template_view v;
    views_of(t, mv, tv, v);
    int ok = check_rolled(ctx, v);
    if (ok != AFIS_OK) return ok;
    append_entry(ctx->hg, v.n_minu > 0 ? &v.minu[0] : nullptr, v.n_tex > 0 ? &v.tex[0] : nullptr);
    return AFIS_OK;
}

// n rolled .dat files at once: parsed on the host's threads (a 100k-file gallery is 5 GB of parsing: 3.7 s on one thread), appended in order.
int afis_gallery_add_dat_batch(afis_ctx* ctx, const void* const* bytes, const size_t* lens, int64_t n, int* load_rc)
{
    if (!ctx || n < 0 || (n > 0 && (!bytes || !lens))) return fail(ctx, AFIS_EINVAL, "afis_gallery_add_dat_batch: bad argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add_dat_batch: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    std::vector<HostTemplate> ts((size_t)n);
    std::vector<int> rcs((size_t)n, 0);
    parallel_for(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            int rc = parse_rolled_dat(bytes[i], lens[i], ts[(size_t)i]);
            if (rc < 0 || rc == 8) { ts[(size_t)i].minu.clear(); ts[(size_t)i].tex.clear(); }     // as afis_gallery_add_dat
            rcs[(size_t)i] = rc;
        }
    });
    std::vector<afis_minutiae_view> mv; std::vector<afis_texture_view> tv; afis_template_view v;
    for (int64_t i = 0; i < n; ++i) {                                      // validate everything before anything is appended
        views_of(ts[(size_t)i], mv, tv, v);
        int ok = check_rolled(ctx, v);
        if (ok != AFIS_OK) return ok;
        if (v.n_tex > 0 && !v.tex[0].codes) return fail(ctx, AFIS_EFORMAT, "afis_gallery_add_dat_batch: rolled texture template without PQ codes");
    }
    // append_entry for all of them at once: the slots follow from the counts, the staged arrays grow once (without a zero-fill) and the templates
    // are copied to their slots by the host's threads (appending one by one was a serial pass over 50 KB per template)
    HostGallery& hg = ctx->hg;
    std::vector<int64_t> mo((size_t)n + 1), to((size_t)n + 1);
    mo[0] = (int64_t)hg.mx.size(); to[0] = (int64_t)hg.tx.size();
    for (int64_t i = 0; i < n; ++i) {
        const HostTemplate& t = ts[(size_t)i];
        mo[(size_t)i + 1] = mo[(size_t)i] + (t.minu.empty() ? 0 : t.minu[0].n());
        to[(size_t)i + 1] = to[(size_t)i] + (t.tex.empty() ? 0 : std::min(t.tex[0].n(), kTexMax));          // matcher.cpp:546-547
    }
    const size_t M = (size_t)mo[(size_t)n], X = (size_t)to[(size_t)n];
    hg.mx.resize(M); hg.my.resize(M); hg.mori.resize(M); hg.mdes.resize(M * kDes);
    hg.tx.resize(X); hg.ty.resize(X); hg.tori.resize(X); hg.tcodes.resize(X * kM);
    parallel_for(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const HostTemplate& t = ts[(size_t)i];
            const size_t a = (size_t)mo[(size_t)i], nm = (size_t)(mo[(size_t)i + 1] - mo[(size_t)i]);
            if (nm) {
                const HostMinutiae& m = t.minu[0];
                memcpy(&hg.mx[a], m.x.data(), nm * 2); memcpy(&hg.my[a], m.y.data(), nm * 2); memcpy(&hg.mori[a], m.ori.data(), nm * 4);
                memcpy(&hg.mdes[a * kDes], m.des.data(), nm * kDes * 4);
            }
            const size_t b = (size_t)to[(size_t)i], nt = (size_t)(to[(size_t)i + 1] - to[(size_t)i]);
            if (nt) {
                const HostTexture& x = t.tex[0];
                memcpy(&hg.tx[b], x.x.data(), nt * 2); memcpy(&hg.ty[b], x.y.data(), nt * 2); memcpy(&hg.tori[b], x.ori.data(), nt * 4);
                memcpy(&hg.tcodes[b * kM], x.codes.data(), nt * kM);
            }
        }
    });
    for (int64_t i = 0; i < n; ++i) {
        hg.minu_off.push_back(mo[(size_t)i + 1]); hg.tex_off.push_back(to[(size_t)i + 1]);
        hg.empty.push_back(mo[(size_t)i + 1] == mo[(size_t)i] && to[(size_t)i + 1] == to[(size_t)i] ? 1 : 0);
        if (load_rc) load_rc[i] = rcs[(size_t)i];
    }
    return AFIS_OK;
}

int afis_gallery_reserve(afis_ctx* ctx, int64_t n_templates)
{
    if (!ctx || n_templates < 0) return fail(ctx, AFIS_EINVAL, "afis_gallery_reserve: bad argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_reserve: gallery already committed");
    if (ctx->pend) return AFIS_OK;                                          // a mapped container is not staged in host arrays at all
    HostGallery& hg = ctx->hg;
    const double have = (double)hg.size();
    if ((double)n_templates <= have) return AFIS_OK;
    const double scale = have > 0 ? (double)n_templates / have * 1.02 : 0;  // 2 % headroom over the running average
    const size_t nm = have > 0 ? (size_t)((double)hg.mx.size() * scale) : (size_t)n_templates * 80;
    const size_t nt = have > 0 ? (size_t)((double)hg.tx.size() * scale) : (size_t)n_templates * 800;
    try {
        hg.mx.reserve(nm); hg.my.reserve(nm); hg.mori.reserve(nm); hg.mdes.reserve(nm * kDes);
        hg.tx.reserve(nt); hg.ty.reserve(nt); hg.tori.reserve(nt); hg.tcodes.reserve(nt * kM);
        hg.minu_off.reserve((size_t)n_templates + 1); hg.tex_off.reserve((size_t)n_templates + 1); hg.empty.reserve((size_t)n_templates);
    } catch (const std::bad_alloc&) { return fail(ctx, AFIS_EINVAL, "afis_gallery_reserve: out of host memory"); }
    return AFIS_OK;
}

int afis_gallery_add_packed(afis_ctx* ctx, int64_t n, const int64_t* minu_off, const int16_t* minu_x, const int16_t* minu_y,
                            const float* minu_ori, const float* minu_des, const int64_t* tex_off, const int16_t* tex_x,
                            const int16_t* tex_y, const float* tex_ori, const uint8_t* tex_codes)
{
    if (!ctx || n < 0 || !minu_off || !tex_off) return fail(ctx, AFIS_EINVAL, "afis_gallery_add_packed: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add_packed: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t nm = minu_off[i + 1] - minu_off[i], nt = tex_off[i + 1] - tex_off[i];
        if (nm < 0 || nm > 2000 || nt < 0 || nt > 2000) return fail(ctx, AFIS_EINVAL, "afis_gallery_add_packed: template point count must be 0..2000");
    }
    HostGallery& hg = ctx->hg;
    const int64_t m0 = minu_off[0], m1 = minu_off[n], t0 = tex_off[0];
    hg.mx.insert(hg.mx.end(), minu_x + m0, minu_x + m1); hg.my.insert(hg.my.end(), minu_y + m0, minu_y + m1);
    hg.mori.insert(hg.mori.end(), minu_ori + m0, minu_ori + m1);
    hg.mdes.insert(hg.mdes.end(), minu_des + m0 * kDes, minu_des + m1 * kDes);
    const int64_t mbase = hg.minu_off.back() - m0;
    for (int64_t i = 0; i < n; ++i) {
        hg.minu_off.push_back(minu_off[i + 1] + mbase);
        const int64_t a = tex_off[i], nt = std::min<int64_t>(tex_off[i + 1] - a, kTexMax);
        hg.tx.insert(hg.tx.end(), tex_x + a, tex_x + a + nt); hg.ty.insert(hg.ty.end(), tex_y + a, tex_y + a + nt);
        hg.tori.insert(hg.tori.end(), tex_ori + a, tex_ori + a + nt);
        hg.tcodes.insert(hg.tcodes.end(), tex_codes + a * kM, tex_codes + (a + nt) * kM);
        hg.tex_off.push_back((int64_t)hg.tx.size());
        hg.empty.push_back((minu_off[i + 1] == minu_off[i] && nt == 0) ? 1 : 0);
    }
    (void)t0;
    return AFIS_OK;
}

// ---- packed gallery container (layout in template_io.h) ---------------------------------------------------
int afis_gallery_save(afis_ctx* ctx, const char* path, const char* const* names)
{
    if (!ctx || !path) return fail(ctx, AFIS_EINVAL, "afis_gallery_save: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_save: the host staging copy is released at commit; save before afis_gallery_commit");
    if (int rc_ = materialise(ctx)) return rc_;
    std::vector<std::string> nm;
    if (names) for (int64_t i = 0; i < ctx->hg.size(); ++i) nm.emplace_back(names[i] ? names[i] : "");
    std::string err;
    if (!write_gallery_container(path, ctx->hg, nm, err)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_save: " + err);
    return AFIS_OK;
}

int afis_gallery_load(afis_ctx* ctx, const char* path, int64_t first, int64_t count)
{
    if (!ctx || !path) return fail(ctx, AFIS_EINVAL, "afis_gallery_load: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_load: gallery already committed");
    std::string err;
    if (ctx->hg.size() == 0 && !ctx->pend) {                               // the usual case (one container, or one shard of it): map it, validate it, read it at the commit
        std::unique_ptr<GalleryMapping> gm = map_gallery_container(path, err);
        if (!gm) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: " + err);
        if (count < 0) count = gm->G - first;
        if (first < 0 || count < 0 || first + count > gm->G) return fail(ctx, AFIS_EFORMAT, std::string("afis_gallery_load: ") + path + ": template range outside the container");
        for (int64_t i = first; i < first + count; ++i) {
            const int64_t nm = gm->minu_off[i + 1] - gm->minu_off[i], nt = gm->tex_off[i + 1] - gm->tex_off[i];
            if (nm > 2000 || nt > kTexMax || (gm->empty[i] != 0) != (nm == 0 && nt == 0)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: template counts out of range");
        }
        ctx->pend = std::move(gm); ctx->pend_first = first; ctx->pend_count = count;
        return AFIS_OK;
    }
    if (int rc_ = materialise(ctx)) return rc_;
    HostGallery add;                                                       // parsed aside so a bad file leaves the staged gallery untouched
    if (!read_gallery_container(path, first, count, add, nullptr, nullptr, err)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: " + err);
    const int64_t n = add.size();
    for (int64_t i = 0; i < n; ++i) {
        const int64_t nm = add.minu_off[i + 1] - add.minu_off[i], nt = add.tex_off[i + 1] - add.tex_off[i];
        if (nm > 2000 || nt > kTexMax || (add.empty[i] != 0) != (nm == 0 && nt == 0)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: template counts out of range");
    }
    HostGallery& hg = ctx->hg;
    if (hg.size() == 0) { hg = std::move(add); return AFIS_OK; }
    const int64_t mb = hg.minu_off.back(), tb = hg.tex_off.back();
    hg.mx.insert(hg.mx.end(), add.mx.begin(), add.mx.end()); hg.my.insert(hg.my.end(), add.my.begin(), add.my.end());
    hg.mori.insert(hg.mori.end(), add.mori.begin(), add.mori.end()); hg.mdes.insert(hg.mdes.end(), add.mdes.begin(), add.mdes.end());
    hg.tx.insert(hg.tx.end(), add.tx.begin(), add.tx.end()); hg.ty.insert(hg.ty.end(), add.ty.begin(), add.ty.end());
    hg.tori.insert(hg.tori.end(), add.tori.begin(), add.tori.end()); hg.tcodes.insert(hg.tcodes.end(), add.tcodes.begin(), add.tcodes.end());
    for (int64_t i = 0; i < n; ++i) { hg.minu_off.push_back(mb + add.minu_off[i + 1]); hg.tex_off.push_back(tb + add.tex_off[i + 1]); hg.empty.push_back(add.empty[i]); }
    return AFIS_OK;
}

int afis_gallery_file_info(const char* path, int64_t* G, int64_t* n_minutiae, int64_t* n_tex_points, int32_t* tex_counts)
{
    if (!path) return fail(nullptr, AFIS_EINVAL, "afis_gallery_file_info: null argument");
    std::string err;
    GalleryFileInfo info;
    if (!gallery_container_info(path, info, err)) return fail(nullptr, AFIS_EFORMAT, "afis_gallery_file_info: " + err);
    if (G) *G = info.G;
    if (n_minutiae) *n_minutiae = info.n_minu;
    if (n_tex_points) *n_tex_points = info.n_tex;
    if (tex_counts) {
        HostGallery none; std::vector<int32_t> tc;
        if (!read_gallery_container(path, 0, 0, none, nullptr, &tc, err, false)) return fail(nullptr, AFIS_EFORMAT, "afis_gallery_file_info: " + err);
        memcpy(tex_counts, tc.data(), tc.size() * sizeof(int32_t));
    }
    return AFIS_OK;
}

int afis_gallery_file_names(const char* path, int64_t first, int64_t count, char* buf, size_t cap, size_t* need)
{
    if (!path || !need) return fail(nullptr, AFIS_EINVAL, "afis_gallery_file_names: null argument");
    std::string err;
    HostGallery none; std::vector<std::string> names;
    if (!read_gallery_container(path, first, count, none, &names, nullptr, err, false)) return fail(nullptr, AFIS_EFORMAT, "afis_gallery_file_names: " + err);
    size_t total = 0;
    for (const std::string& n : names) total += n.size() + 1;
    *need = total;
    if (!buf) return AFIS_OK;
    if (cap < total) return fail(nullptr, AFIS_EINVAL, "afis_gallery_file_names: buffer too small");
    char* w = buf;
    for (const std::string& n : names) { memcpy(w, n.c_str(), n.size() + 1); w += n.size() + 1; }
    return AFIS_OK;
}

// The arrays of a shard are 50 KB per template (5 GB per 100 000): a pageable hipMemcpy moves them at 8-11 GB/s through the runtime's one staging thread.
// Here they go through two pinned 64 MB buffers: the host's threads fill one (from the staged arrays or straight from a mapped container: that is where
// the page cache is read) while the DMA engine empties the other.
struct PinnedPipe {
    static constexpr size_t kCap = (size_t)64 << 20;
    void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; int k = 0;
    hipError_t init()
    {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&buf[i], kCap, hipHostMallocDefault); if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    ~PinnedPipe() { for (int i = 0; i < 2; ++i) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (buf[i]) (void)hipHostFree(buf[i]); } }
};

static hipError_t upload_bulk(PinnedPipe& pp, DevBuf& b, const void* src, size_t bytes, hipStream_t s)
{
    hipError_t e = b.ensure(std::max<size_t>(bytes, 16));
    if (e != hipSuccess || bytes == 0) return e;
    if (bytes < ((size_t)4 << 20)) return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s);
    for (size_t off = 0; off < bytes; off += PinnedPipe::kCap) {
        const size_t n = std::min(PinnedPipe::kCap, bytes - off);
        const int slot = pp.k & 1;
        if (pp.used[slot]) { e = hipEventSynchronize(pp.ev[slot]); if (e != hipSuccess) return e; }
        const uint8_t* from = (const uint8_t*)src + off; uint8_t* to = (uint8_t*)pp.buf[slot];
        parallel_for((int64_t)((n + 4095) / 4096), [&](int64_t lo, int64_t hi) { const size_t a = (size_t)lo * 4096, z = std::min(n, (size_t)hi * 4096); memcpy(to + a, from + a, z - a); });
        e = hipMemcpyAsync((uint8_t*)b.p + off, pp.buf[slot], n, hipMemcpyHostToDevice, s); if (e != hipSuccess) return e;
        e = hipEventRecord(pp.ev[slot], s); if (e != hipSuccess) return e;
        pp.used[slot] = true; ++pp.k;
    }
    return hipSuccess;
}

static int ensure_mf_gallery(afis_ctx* ctx, hipStream_t s);

int afis_gallery_commit(afis_ctx* ctx, int64_t index_base)
{
    if (!ctx) return AFIS_EINVAL;
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_commit: already committed");
    // The staged shard as plain arrays: ctx->hg, or the mapped container's range (offsets rebased to the shard's first point).
    HostGallery& hg = ctx->hg;
    const GalleryMapping* gm = ctx->pend.get();
    const int64_t G = gm ? ctx->pend_count : hg.size();
    const int64_t* src_mo = gm ? gm->minu_off + ctx->pend_first : hg.minu_off.data();
    const int64_t* src_to = gm ? gm->tex_off + ctx->pend_first : hg.tex_off.data();
    const int64_t m0 = src_mo[0], t0 = src_to[0];
    const size_t NM = (size_t)(src_mo[G] - m0), NT = (size_t)(src_to[G] - t0);
    const int16_t* s_mx = gm ? gm->mx + m0 : hg.mx.data(); const int16_t* s_my = gm ? gm->my + m0 : hg.my.data();
    const float* s_mori = gm ? gm->mori + m0 : hg.mori.data(); const float* s_mdes = gm ? gm->mdes + (size_t)m0 * kDes : hg.mdes.data();
    const int16_t* s_tx = gm ? gm->tx + t0 : hg.tx.data(); const int16_t* s_ty = gm ? gm->ty + t0 : hg.ty.data();
    const float* s_tori = gm ? gm->tori + t0 : hg.tori.data(); const uint8_t* s_tcodes = gm ? gm->tcodes + (size_t)t0 * kM : hg.tcodes.data();
    const uint8_t* s_empty = gm ? gm->empty + ctx->pend_first : hg.empty.data();
    if (G > 0x7fffffff / 8 || NM > 0x7fffffffull || NT > 0x7fffffffull)
        return fail(ctx, AFIS_EINVAL, "afis_gallery_commit: shard too large for 32-bit point offsets; split the gallery into more shards");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool clock_it = getenv("AFIS_COMMIT_TIMING") != nullptr;           // development aid: where the commit's time goes, on stderr
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) { if (clock_it) { (void)hipStreamSynchronize(ctx->stream); const double t = now(); fprintf(stderr, "commit: %-28s %8.1f ms\n", what, t - t_prev); t_prev = t; } };
    PinnedPipe pp;
    HIPCHK(ctx, pp.init());
    lap("pinned buffers");
    std::vector<int32_t> mo(G + 1), to(G + 1);
    int max_nR = 0;
    for (int64_t i = 0; i <= G; ++i) { mo[i] = (int32_t)(src_mo[i] - m0); to[i] = (int32_t)(src_to[i] - t0); }
    for (int64_t i = 0; i < G; ++i) max_nR = std::max(max_nR, mo[i + 1] - mo[i]);
    HIPCHK(ctx, upload_bulk(pp, ctx->g_minu_des, s_mdes, NM * kDes * sizeof(float), ctx->stream));     // the big one first: the fragment kernel below runs while the rest is uploaded
    lap("minutiae descriptors");
    HIPCHK(ctx, upload(ctx->g_minu_off, mo, ctx->stream));
    std::vector<int32_t> toff((size_t)G + 1, 0);
    {   // the descriptors as MFMA operand fragments: laid out on the device from the descriptors just uploaded (round 3 transposed them on the host and uploaded another 34 KB per template)
        for (int64_t t = 0; t < G; ++t) toff[(size_t)t + 1] = toff[(size_t)t] + (mo[t + 1] - mo[t] + 15) / 16;
        HIPCHK(ctx, upload(ctx->g_minu_tile_off, toff, ctx->stream));
        HIPCHK(ctx, ctx->g_minu_frag.ensure(std::max<size_t>((size_t)toff[(size_t)G] * 6 * 64 * 16, 16)));
        HIPCHK(ctx, launch_fragment_tiles(ctx->g_minu_des.as<float>(), ctx->g_minu_off.as<int32_t>(), ctx->g_minu_tile_off.as<int32_t>(), (int)G, ctx->g_minu_frag.p, ctx->stream));
    }
    lap("fragment tiles");
    std::vector<short2> mxy(NM), txy(NT);
    parallel_for((int64_t)NM, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) mxy[(size_t)i] = make_short2(s_mx[i], s_my[i]); });
    parallel_for((int64_t)NT, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) txy[(size_t)i] = make_short2(s_tx[i], s_ty[i]); });
    lap("xy packing");
    HIPCHK(ctx, upload_bulk(pp, ctx->g_minu_xy, mxy.data(), NM * sizeof(short2), ctx->stream));
    HIPCHK(ctx, upload_bulk(pp, ctx->g_minu_ori, s_mori, NM * sizeof(float), ctx->stream));
    HIPCHK(ctx, upload(ctx->g_tex_off, to, ctx->stream));
    HIPCHK(ctx, upload_bulk(pp, ctx->g_tex_xy, txy.data(), NT * sizeof(short2), ctx->stream));
    HIPCHK(ctx, upload_bulk(pp, ctx->g_tex_ori, s_tori, NT * sizeof(float), ctx->stream));
    lap("small arrays");
    HIPCHK(ctx, upload_bulk(pp, ctx->g_tex_codes, s_tcodes, NT * kM, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    lap("texture codes");
    {   // block offsets of the direct conflict-free kernel's code stream (variants 6 / 7): (blocks + 1) x 64 entries per template.  The stream
        // itself — a full copy of the PQ codes — is laid out on the device at the first use of those variants (k_codes_cf); the default path
        // never builds it.
        std::vector<int32_t> cfb(G + 1);
        int64_t nblk = 0;
        for (int64_t t = 0; t < G; ++t) { cfb[t] = (int32_t)nblk; const int64_t n = (int64_t)(to[t + 1] - to[t]); nblk += n > 0 ? (n + 63) / 64 + 1 : 0; }
        cfb[G] = (int32_t)nblk;
        if (nblk > 0x7fffffff / 64) return fail(ctx, AFIS_EINVAL, "afis_gallery_commit: shard too large for the ADC code stream; split the gallery into more shards");
        ctx->cf_blocks = nblk; ctx->codes_cf_built = false;
        HIPCHK(ctx, upload(ctx->g_tex_cf_blk, cfb, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    {   // block offsets of the quantised path's code stream (ceil(n/64) blocks per template); the stream itself is made on first use
        std::vector<int32_t> qb(G + 1);
        int64_t nb = 0;
        for (int64_t t = 0; t < G; ++t) { qb[t] = (int32_t)nb; nb += ((int64_t)(to[t + 1] - to[t]) + 63) / 64; }
        qb[G] = (int32_t)nb;
        ctx->q_blocks = nb;
        HIPCHK(ctx, upload(ctx->g_tex_q_blk, qb, ctx->stream));
    }
    {   // tile offsets of the matrix-core bound pass's stream (ceil(n/32) tiles of 32 points per template); the stream itself is made on first use
        std::vector<int32_t> tb(G + 1);
        int64_t nt = 0;
        for (int64_t t = 0; t < G; ++t) { tb[t] = (int32_t)nt; nt += ((int64_t)(to[t + 1] - to[t]) + 31) / 32; }
        tb[G] = (int32_t)nt;
        if (nt > 0x7fffffff / 32) return fail(ctx, AFIS_EINVAL, "afis_gallery_commit: shard too large for the bound pass's code stream; split the gallery into more shards");
        ctx->t32_tiles = nt;
        HIPCHK(ctx, upload(ctx->g_tex_t32_blk, tb, ctx->stream));
    }
    { DevBuf& eb = ctx->g_empty; HIPCHK(ctx, eb.ensure(std::max<size_t>((size_t)G, 16))); if (G) HIPCHK(ctx, hipMemcpyAsync(eb.p, s_empty, (size_t)G, hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(ctx, ctx->g_task_ctr.ensure(64));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    GalleryDev& g = ctx->gal;
    g.G = (int32_t)G;
    g.minu_off = ctx->g_minu_off.as<int32_t>(); g.minu_xy = ctx->g_minu_xy.as<short2>(); g.minu_ori = ctx->g_minu_ori.as<float>();
    g.minu_des = ctx->g_minu_des.as<float>(); g.minu_frag = ctx->g_minu_frag.as<float4>(); g.minu_tile_off = ctx->g_minu_tile_off.as<int32_t>(); g.tex_off = ctx->g_tex_off.as<int32_t>(); g.tex_xy = ctx->g_tex_xy.as<short2>();
    g.tex_ori = ctx->g_tex_ori.as<float>(); g.tex_codes = ctx->g_tex_codes.as<uint4>(); g.tex_codes_cf = nullptr; g.tex_cf_blk = ctx->g_tex_cf_blk.as<int32_t>(); g.empty = ctx->g_empty.as<uint8_t>();
    g.task_ctr = ctx->g_task_ctr.as<int32_t>();
    ctx->max_nR = max_nR;
    ctx->total_tex_points = (int64_t)NT; ctx->total_minutiae = (int64_t)NM;
    ctx->index_base = index_base;
    ctx->committed = true;
    if (ctx->adc_variant == 9 && G > 0) {                                    // the default path's derived streams belong to the resident gallery: built here, not by the first search
        int rcg = ensure_mf_gallery(ctx, ctx->stream);
        if (rcg != AFIS_OK) { ctx->committed = false; return rcg; }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        lap("bound pass's code stream");
    }
    // the host staging copy is no longer needed
    std::vector<uint8_t> e(s_empty, s_empty + G);
    if (hg.mdes.capacity() > ((size_t)16 << 20)) {                           // a large staging copy is released by a thread of its own
        // The pages go back in 32 MB pieces (madvise takes the address-space lock shared and briefly); one munmap of 3 GB holds it exclusively for
        // a third of a second, and every allocation the caller makes next — the commit's own clean-up, the first search — would wait for it.
        HostGallery* old = new HostGallery(std::move(ctx->hg));
        ctx->staging_reaper = std::thread([old]() {
            std::vector<std::pair<uintptr_t, size_t>> pieces;
            auto drop = [&](void* p, size_t bytes) {
                const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, z = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
                for (uintptr_t q = a; q < z; q += (uintptr_t)32 << 20) pieces.emplace_back(q, (size_t)std::min<uintptr_t>((uintptr_t)32 << 20, z - q));
            };
            drop(old->mdes.data(), old->mdes.capacity() * sizeof(float)); drop(old->tcodes.data(), old->tcodes.capacity());
            drop(old->mori.data(), old->mori.capacity() * 4); drop(old->tori.data(), old->tori.capacity() * 4);
            std::atomic<size_t> next{0};
            auto work = [&]() { for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) (void)madvise((void*)pieces[i].first, pieces[i].second, MADV_DONTNEED); };
            std::thread helpers[3];                                          // four threads return 5 GB in a quarter of the time one takes
            for (std::thread& h : helpers) h = std::thread(work);
            work();
            for (std::thread& h : helpers) h.join();
            delete old;
        });
    }
    lap("offset tables");
    ctx->hg = HostGallery(); ctx->hg.empty = std::move(e);
    ctx->pend.reset(); ctx->pend_first = ctx->pend_count = 0;
    lap("staging released");
    return AFIS_OK;
}

int64_t afis_gallery_size(const afis_ctx* ctx) { return !ctx ? 0 : ctx->pend ? ctx->pend_count : (int64_t)ctx->hg.empty.size(); }

// ---------------------------------------------------------------------------------------------------------------------
static const int kSelected[3] = {27 - 1, 3 - 1, 12 - 1};                   // matcher.cpp:380

// spec == NULL: the reference's selection for every query (templates 27, 3, 12 and texture template 0, matcher.cpp:380-415).
// spec != NULL (afis_match_all_templates): query i uses latent minutiae templates spec[i*4 + 0..2] (-1 = none) and latent texture
// template spec[i*4 + 3] (-1 = none), and is never "latent empty".
static int build_group(afis_ctx* ctx, const afis_template_view* qs, int nq, QueryGroup& grp, std::vector<int32_t>& status_out, const int* spec = nullptr)
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

int afis_queries_upload(afis_ctx* ctx, const afis_template_view* queries, int n_q, afis_queries** out)
{
    if (!ctx || !out || n_q < 0 || (n_q > 0 && !queries)) return fail(ctx, AFIS_EINVAL, "afis_queries_upload: bad argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_queries_upload: commit the gallery first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // group size: bounded by the option and by the memory budget of a group's per-pair buffers
    const int64_t G = std::max<int64_t>(1, ctx->gal.G);
    const int64_t by_mem = group_budget_bytes(ctx) / group_bytes_per_query(ctx, G);
    // latents per launch group: the option, or (0 = auto) as many as keep about two million (latent, rolled) pairs in a launch — 20 at a 100k-template
    // shard, 128 at <= 15k (a 12.5k-template shard: 100 latents in one launch 310.7 ms, in 64 + 36: 315.1): the persistent per-pair kernels lose their tails once per launch, which shows on small shards (12 launches of 100k pairs
    // each cost 1.2 x their share of a 100k-template step; 2 launches do not).  Measured at 100k templates, 100 latents: 7 per launch 2 495 ms, 10: 2 486,
    // 15: 2 466, 20: 2 463, 34: 2 468.
    const int64_t want = ctx->query_batch > 0 ? ctx->query_batch : std::min<int64_t>(128, std::max<int64_t>(10, (2000000 + G / 2) / G));
    int per = (int)std::max<int64_t>(1, std::min<int64_t>(want, by_mem));
    afis_queries* q = new afis_queries();
    q->n_q = n_q;
    // Launch groups are contiguous runs of at most `per` queries.  The matrix-core bound pass (adc_variant 9) works in row groups of 768 latent
    // texture rows: a run whose rows fill its last row group only partly pays for the whole of it, so the cuts are placed where the total
    // number of row groups is smallest (dynamic programme over the cut positions; ties: fewer launches).  Results do not depend on the cuts.
    std::vector<int> cuts;                                                  // group ends (exclusive)
    if (ctx->adc_variant == 9 && n_q > 1) {
        std::vector<long long> rows((size_t)n_q + 1, 0);
        for (int i = 0; i < n_q; ++i) {
            const afis_template_view& t = queries[i];
            const bool has = t.n_tex > 0 && t.tex && !(t.n_minu <= kSelected[0] && t.n_tex <= 0);
            rows[(size_t)i + 1] = rows[(size_t)i] + (has ? std::min(std::max(t.tex[0].n, 0), kTexMax) : 0);
        }
        const long long kInf = 1ll << 60;
        const long long rg_rows = ctx->mf_blocks == 102 ? 512 : 768;
        std::vector<long long> best((size_t)n_q + 1, kInf); std::vector<int> from((size_t)n_q + 1, 0), cnt((size_t)n_q + 1, 0);
        best[0] = 0;
        for (int i = 1; i <= n_q; ++i)
            for (int j = std::max(0, i - per); j < i; ++j) {
                const long long c = best[(size_t)j] + (rows[(size_t)i] - rows[(size_t)j] + rg_rows - 1) / rg_rows;
                if (c < best[(size_t)i] || (c == best[(size_t)i] && cnt[(size_t)j] + 1 < cnt[(size_t)i])) { best[(size_t)i] = c; from[(size_t)i] = j; cnt[(size_t)i] = cnt[(size_t)j] + 1; }
            }
        for (int i = n_q; i > 0; i = from[(size_t)i]) cuts.push_back(i);
        std::reverse(cuts.begin(), cuts.end());
    } else {
        for (int i = per; i < n_q; i += per) cuts.push_back(i);
        if (n_q > 0) cuts.push_back(n_q);
    }
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
    for (QueryGroup& g : q->groups) g.release();
    delete q;
}

// variants 6 / 7 read the gallery's codes from their own lane-ordered stream: lay it out now if this is their first use
static int ensure_codes_cf(afis_ctx* ctx, int variant)
{
    if ((variant != 6 && variant != 7) || ctx->codes_cf_built) return AFIS_OK;
    HIPCHK(ctx, ctx->g_tex_codes_cf.ensure(std::max<size_t>((size_t)ctx->cf_blocks * 64 * 16, 16)));
    ctx->gal.tex_codes_cf = ctx->g_tex_codes_cf.as<uint4>();
    HIPCHK(ctx, launch_codes_cf(ctx->gal, ctx->g_tex_codes_cf.p, ctx->stream));
    ctx->codes_cf_built = true;
    return AFIS_OK;
}

static int tile_share_of(const afis_ctx* ctx) { return ctx->tile_share > 0 ? ctx->tile_share : 4; }

// S4 + S5 + S6 of adc_variant 8 for one query group (rm_val / rm_arg sized by the caller): the quantised pass bounds the candidates, the fp32
// table (reference layout, all rows of the group) settles them
static int adc_stage_q(afis_ctx* ctx, QueryGroup& grp, int chunk, bool exact, hipEvent_t after_lut = nullptr)
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
static int ensure_mf_gallery(afis_ctx* ctx, hipStream_t s)
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
static int adc_refine_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, bool compact);
static int adc_stage_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, hipEvent_t after_lut = nullptr, hipEvent_t after_bound = nullptr, bool compact = false, hipStream_t sb = nullptr, bool refine_now = true, unsigned long long* diag = nullptr)
{
    const QueryDev& d = grp.dev;
    hipStream_t s = sb ? sb : ctx->stream;
    const GalleryDev& g = ctx->gal;
    if (grp.n_lt_rows <= 0 || g.G <= 0) { if (after_lut) HIPCHK(ctx, hipEventRecord(after_lut, s)); if (after_bound) HIPCHK(ctx, hipEventRecord(after_bound, s)); return AFIS_OK; }
    { int rcg = ensure_mf_gallery(ctx, s); if (rcg != AFIS_OK) return rcg; }
    const int n_rows = grp.n_lt_rows, n_rb = (n_rows + 31) / 32, R_pad = n_rb * 32;
    // The per-row buffers are sized for the group's WORST case (every latent with kTexMax rows), as afis_search_resident has already done before queuing anything: these
    // calls find them large enough (a hipMalloc behind queued work was seen to take 0.5-0.8 s; see there).  Callers outside a search (the parity taps) allocate here.
    const size_t R_cap = std::max<size_t>((size_t)R_pad, ((size_t)grp.nq * kTexMax + 31) / 32 * 32);
    HIPCHK(ctx, ctx->mf_bfrag.ensure(R_cap / 32 * 6 * 64 * 16));
    HIPCHK(ctx, ctx->mf_rowk.ensure(R_cap * 16));
    HIPCHK(ctx, ctx->mf_rec.ensure((size_t)g.G * R_cap * kMfRecBytesPerRow));
    if (ctx->mf_collect_stats && !ctx->mf_stats.p) { HIPCHK(ctx, ctx->mf_stats.ensure(64)); HIPCHK(ctx, hipMemsetAsync(ctx->mf_stats.p, 0, 64, s)); }
    HIPCHK(ctx, launch_mf_rows(d.lt_des, n_rows, n_rb, ctx->codewords.as<float>(), ctx->mf_cwn.as<float>(), ctx->mf_bfrag.p, ctx->mf_rowk.p, s));
    if (after_lut) HIPCHK(ctx, hipEventRecord(after_lut, s));
    // workgroups = row groups x gallery chunks: about 24 per CU (a CU runs one at a time: the end of the launch idles at most ~1/24 of it),
    // a chunk never below 8 templates
    const int wg_rb = ctx->mf_blocks == 102 ? 16 : 24;                 // row blocks per workgroup (adc_mfma.hip)
    const int n_rg = (n_rb + wg_rb - 1) / wg_rb;
    const long long want_chunks = std::max<long long>(1, (256 * 24) / n_rg);
    const int chunk = ctx->chunk > 0 ? ctx->chunk : (int)std::max<long long>(8, ((long long)g.G + want_chunks - 1) / want_chunks);
    HIPCHK(ctx, launch_adc_mfma(g, ctx->g_codes_p.p, ctx->g_nrm_p.as<float>(), ctx->g_tile_meta.p, ctx->g_tex_t32_blk.as<int32_t>(), ctx->mf_cw16.p,
                                ctx->mf_bfrag.p, ctx->mf_rowk.p, n_rows, n_rb, R_pad, chunk, ctx->mf_blocks, ctx->mf_rec.p, diag, s));
    if (after_bound) HIPCHK(ctx, hipEventRecord(after_bound, s));
    return refine_now ? adc_refine_mfma(ctx, grp, all_rows, compact) : AFIS_OK;
}

static int adc_refine_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, bool compact)
{
    if (grp.n_lt_rows <= 0 || ctx->gal.G <= 0) return AFIS_OK;
    const int R_pad = (grp.n_lt_rows + 31) / 32 * 32;
    HIPCHK(ctx, launch_tex_refine(grp.dev, ctx->gal, ctx->codewords.as<float>(), ctx->mf_rec.p, ctx->mf_rowk.p, R_pad, all_rows ? 1 : 0, ctx->rm_val.as<float>(),
                                  ctx->rm_arg.as<int32_t>(), ctx->mf_collect_stats ? ctx->mf_stats.as<unsigned long long>() : nullptr,
                                  compact ? ctx->rm_cv.as<float>() : nullptr, compact ? ctx->rm_n.as<int32_t>() : nullptr, ctx->stream));
    return AFIS_OK;
}

// Rank lists are made on the device for k <= kDeviceTopK (k passes of a workgroup-wide maximum per query); larger k sorts on the host.
static const int kDeviceTopK = 64;

int afis_search_resident(afis_ctx* ctx, afis_queries* q, float* scores, float* parts, int32_t* status,
                         int k, int64_t* topk_idx, float* topk_score)
{
    if (!ctx || !q) return fail(ctx, AFIS_EINVAL, "afis_search_resident: null argument");
    if (!ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_search: commit the gallery first");
    if (k < 0 || (k > 0 && (!topk_idx || !topk_score))) return fail(ctx, AFIS_EINVAL, "afis_search: k > 0 needs topk_idx and topk_score");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const GalleryDev& g = ctx->gal;
    const int64_t G = g.G;
    const int nq_all = q->n_q;
    afis_timing tm = {};
    if (status) for (int i = 0; i < nq_all; ++i) status[i] = q->status[i];
    hipStream_t s = ctx->stream;
    // The groups run back to back on the stream (the overlapped schedule adds one host round trip per group: the wait for its side streams).  Scores of ALL queries stay on the device
    // ([n_q][G]) for the rank-list kernel; they cross PCIe only when the caller asks for them.
    const size_t n_groups = q->groups.size();
    while (ctx->evpool.size() < n_groups * 10 + 2) { hipEvent_t e; HIPCHK(ctx, hipEventCreate(&e)); ctx->evpool.push_back(e); }
    if (G > 0 && nq_all > 0) HIPCHK(ctx, ctx->scores.ensure((size_t)nq_all * G * 4));
    HIPCHK(ctx, ctx->diag.ensure(std::max<size_t>(n_groups, 1) * kDiagWords * 8));
    HIPCHK(ctx, hipMemsetAsync(ctx->diag.p, 0, std::max<size_t>(n_groups, 1) * kDiagWords * 8, s));    // before the first group's ev[0]: ordered before everything the side streams do
    // Every buffer of the launch groups is brought to its size HERE, while the device is idle and before anything of this search is queued: for the largest group of
    // the search and for its worst case (every latent with kTexMax texture rows — what group_bytes_per_query budgets), so that the calls further down never
    // re-allocate.  A hipMalloc of 6-13 GB takes 0.3 ms on an idle device; issued behind queued work (the row records used to be allocated inside adc_stage_mfma, after
    // the group's first kernels) it took 510-790 ms in three runs of ten (match -ldir: one search call in seven; profiles/r04_alloc_trace.txt).
    if (G > 0) {
        int nq_max = 0, nL_max = 1, lt_pad_max = 0;
        for (const QueryGroup& grp : q->groups) { nq_max = std::max(nq_max, grp.nq); nL_max = std::max(nL_max, grp.max_nL); lt_pad_max = std::max(lt_pad_max, grp.dev.lt_pad); }
        const size_t n_pairs = (size_t)nq_max * G;
        const size_t lt_cap = std::max<size_t>((size_t)lt_pad_max, ((size_t)kTexMax + kTileRows - 1) / kTileRows * kTileRows);
        if (n_pairs > 0) {
            HIPCHK(ctx, ctx->rm_val.ensure(n_pairs * lt_cap * 4));
            HIPCHK(ctx, ctx->rm_arg.ensure(n_pairs * lt_cap * 4));
            HIPCHK(ctx, ctx->parts.ensure(n_pairs * 16));
            HIPCHK(ctx, ctx->cands.ensure(n_pairs * 3 * kTopMinu * sizeof(MinuCand)));
            HIPCHK(ctx, ctx->cand_n.ensure(n_pairs * 3 * 4));
            HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(n_pairs * 3, (size_t)G) * 4));
            {   // the generic candidate kernel's scratch (sized as in the loop below, for the longest latent minutiae template of the search)
                const size_t per_wg = 2 * (((size_t)nL_max * std::max(1, ctx->max_nR) + 63) / 64 * 64) + 4096;
                int n_wg = 1024;
                while (n_wg > 64 && per_wg * 4 * n_wg > (8ull << 30)) n_wg /= 2;
                HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * n_wg));
            }
            if (ctx->adc_variant == 9) {
                HIPCHK(ctx, ctx->rm_cv.ensure(n_pairs * lt_cap * 4)); HIPCHK(ctx, ctx->rm_n.ensure(n_pairs * 4));
                const size_t R_cap = ((size_t)nq_max * kTexMax + 31) / 32 * 32;
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
    int q0 = 0;
    size_t gi = 0;
    SideStreamGuard side_guard(ctx);
    for (QueryGroup& grp : q->groups) {
        const QueryDev& d = grp.dev;
        const int nq = grp.nq;
        hipEvent_t* ev = &ctx->evpool[gi * 10];
        unsigned long long* const diag_row = ctx->diag.as<unsigned long long>() + gi * kDiagWords;
        if (G > 0) {
            const size_t n_pairs = (size_t)nq * G;
            if (ctx->adc_variant < 8) HIPCHK(ctx, ctx->lut.ensure(std::max<size_t>((size_t)d.n_tiles * kTileFloats * 4, 16)));   // tile LUT of the direct kernels only
            const size_t lt_cap = std::max<size_t>((size_t)d.lt_pad, ((size_t)kTexMax + kTileRows - 1) / kTileRows * kTileRows);    // worst case, as budgeted: no re-allocation when a later group's longest latent is longer (adc_stage_mfma)
            HIPCHK(ctx, ctx->rm_val.ensure(std::max<size_t>(n_pairs * lt_cap * 4, 16)));
            HIPCHK(ctx, ctx->rm_arg.ensure(std::max<size_t>(n_pairs * lt_cap * 4, 16)));
            if (ctx->adc_variant == 9) { HIPCHK(ctx, ctx->rm_cv.ensure(std::max<size_t>(n_pairs * lt_cap * 4, 16))); HIPCHK(ctx, ctx->rm_n.ensure(std::max<size_t>(n_pairs * 4, 16))); }
            HIPCHK(ctx, ctx->parts.ensure(n_pairs * 16));
            // minutiae scratch per workgroup: simi[n] | keys[n] | rowsum[2048] | colsum[2048]  (only pairs the fast kernel cannot take use it)
            size_t per_wg = 2 * (((size_t)std::max(1, grp.max_nL) * std::max(1, ctx->max_nR) + 63) / 64 * 64) + 4096;
            int n_wg = 1024;
            while (n_wg > 64 && per_wg * 4 * n_wg > (8ull << 30)) n_wg /= 2;
            HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * n_wg));
            HIPCHK(ctx, ctx->cands.ensure(n_pairs * 3 * kTopMinu * sizeof(MinuCand)));
            HIPCHK(ctx, ctx->cand_n.ensure(n_pairs * 3 * 4));
            HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(n_pairs * 3, (size_t)G) * 4));
            float* grp_scores = ctx->scores.as<float>() + (size_t)q0 * G;
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
                HIPCHK(ctx, launch_minu_cands(d, g, ctx->scratch.as<float>(), per_wg, n_wg, ctx->minu_generic, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, diag_row, s));
                HIPCHK(ctx, hipEventRecord(ev[7], s));
                HIPCHK(ctx, launch_graph_minutiae(d, g, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->parts.as<float>(), nullptr, nullptr, nullptr, nullptr, 2, s));
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
                HIPCHK(ctx, launch_minu_cands(d, g, ctx->scratch.as<float>(), per_wg, n_wg, ctx->minu_generic, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, diag_row, sh));
                HIPCHK(ctx, hipMemsetAsync(g.task_ctr + 1, 0, 4, sh));                     // the list counter both instances of the list kernel draw from: reset BEFORE either may start
                HIPCHK(ctx, hipEventRecord(ev[7], sh));
                HIPCHK(ctx, launch_graph_minutiae(d, g, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->parts.as<float>(), nullptr, nullptr, nullptr, nullptr, 2, sh, true));
                HIPCHK(ctx, hipEventRecord(ev[4], sh));
                HIPCHK(ctx, hipStreamWaitEvent(s, ev[6], 0));
                HIPCHK(ctx, hipEventRecord(ev[8], s));                                     // the bound pass is done
                rc9 = adc_refine_mfma(ctx, grp, false, true);
                if (rc9 != AFIS_OK) return rc9;
                HIPCHK(ctx, hipEventRecord(ev[2], s));
                HIPCHK(ctx, launch_graph_texture(d, g, ctx->table.as<float>(), ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), ctx->rm_cv.as<float>(), ctx->rm_n.as<int32_t>(), ctx->parts.as<float>(), nullptr, nullptr, 2, s));
                HIPCHK(ctx, hipEventRecord(ev[3], s));
                HIPCHK(ctx, hipStreamWaitEvent(s, ev[7], 0));                              // every candidate list exists: help with whatever lists are left
                HIPCHK(ctx, launch_graph_minutiae(d, g, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->parts.as<float>(), nullptr, nullptr, nullptr, nullptr, 2, s, true));
                HIPCHK(ctx, hipStreamWaitEvent(s, ev[4], 0));
                // The host waits for the two side streams here — one host round trip per launch group; the context's stream has its whole share of the group queued and
                // keeps the chip busy meanwhile.  Without it the run hangs: waiting on the context's stream alone — or on an event of a side stream — never returns although
                // every stream drains at once when it is waited for itself (ROCm 7.2; tools/repro/side_stream_hang.hip is the minimal form).  The wait is bounded.
                { const int rcw = wait_streams(ctx, {sl, sh}, "afis_search: side streams of a launch group"); side_guard.disarm(); if (rcw != AFIS_OK) return rcw; }
            } else {
            if (ctx->adc_variant == 9) {                                    // fp16 matrix-core bound pass + exact recomputation
                int rc9 = adc_stage_mfma(ctx, grp, false, ev[1], ev[6], true, nullptr, true, diag_row);
                if (rc9 != AFIS_OK) return rc9;
            } else if (ctx->adc_variant == 8) {                             // 16-bit fixed-point LDS-table bound pass + exact refine
                int rc16 = adc_stage_q(ctx, grp, chunk, true, ev[1]);
                if (rc16 != AFIS_OK) return rc16;
            } else {
                { int rcf = ensure_codes_cf(ctx, ctx->adc_variant); if (rcf != AFIS_OK) return rcf; }
                HIPCHK(ctx, launch_lut_build(d, ctx->codewords.as<float>(), ctx->lut.as<float>(), ctx->adc_variant, s));
                HIPCHK(ctx, hipEventRecord(ev[1], s));
                HIPCHK(ctx, launch_adc_rowmax(d, g, ctx->lut.as<float>(), chunk, ctx->adc_variant, ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), s));
            }
            HIPCHK(ctx, hipEventRecord(ev[2], s));
            HIPCHK(ctx, launch_graph_texture(d, g, ctx->table.as<float>(), ctx->rm_val.as<float>(), ctx->rm_arg.as<int32_t>(), compact9 ? ctx->rm_cv.as<float>() : nullptr,
                                             compact9 ? ctx->rm_n.as<int32_t>() : nullptr, ctx->parts.as<float>(), nullptr, nullptr, 2, s));
            HIPCHK(ctx, hipEventRecord(ev[3], s));
            { int rcm = minutiae_stage(); if (rcm != AFIS_OK) return rcm; }
            HIPCHK(ctx, hipEventRecord(ev[4], s));
            }
            HIPCHK(ctx, hipEventRecord(ev[9], s));
            HIPCHK(ctx, launch_fuse(d, g, ctx->parts.as<float>(), grp_scores, s));
            HIPCHK(ctx, hipEventRecord(ev[5], s));
            // per-part scores only on request (tests, the all-templates mode); stream order keeps the buffer intact until the copy is done
            if (parts) HIPCHK(ctx, hipMemcpyAsync(parts + (size_t)q0 * G * 4, ctx->parts.p, n_pairs * 16, hipMemcpyDeviceToHost, s));
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
        HIPCHK(ctx, hipMemcpyAsync(topk_idx, ctx->topk_idx.p, (size_t)nq_all * k * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(topk_score, ctx->topk_score.p, (size_t)nq_all * k * 4, hipMemcpyDeviceToHost, s));
    }
    const bool host_topk = k > 0 && !dev_topk;
    float* h_sc = scores;
    if (G > 0 && nq_all > 0 && (scores || host_topk)) {
        if (!h_sc) { ctx->h_scores.resize((size_t)nq_all * G); h_sc = ctx->h_scores.data(); }
        HIPCHK(ctx, hipMemcpyAsync(h_sc, ctx->scores.p, (size_t)nq_all * G * 4, hipMemcpyDeviceToHost, s));
    }
    ctx->h_diag.assign(std::max<size_t>(n_groups, 1) * kDiagWords, 0ull);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_diag.data(), ctx->diag.p, ctx->h_diag.size() * 8, hipMemcpyDeviceToHost, s));
    { const int rcw = wait_streams(ctx, {s}, "afis_search"); if (rcw != AFIS_OK) return rcw; }
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
        const size_t per_wg = 2 * (((size_t)std::max(1, grp.max_nL) * std::max(1, ctx->max_nR) + 63) / 64 * 64) + 4096;
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
            if (launch_minu_cands(grp.dev, one, ctx->scratch.as<float>(), per_wg, n_wg, ctx->minu_generic, cands, cand_n, ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, nullptr, s) != hipSuccess ||
                launch_graph_minutiae(grp.dev, one, cands, cand_n, ctx->parts.as<float>() + (size_t)i * 4,
                                      d_xy.as<short4>() + (size_t)i * 3 * kTopMinu, d_n.as<int32_t>() + (size_t)i * 3, nullptr, nullptr, 2, s) != hipSuccess)
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
        q.groups.back().release();
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

// The round-2 header's struct ended at `pairs` (48 bytes); a caller compiled against it must not be written past that.
int afis_get_timing(const afis_ctx* ctx, afis_timing* out)
{
    return afis_get_timing2(ctx, out, offsetof(afis_timing, pairs) + sizeof(int64_t));
}

int afis_get_timing2(const afis_ctx* ctx, afis_timing* out, size_t struct_size)
{
    if (!ctx || !out || struct_size < sizeof(float)) return AFIS_EINVAL;
    memcpy(out, &ctx->timing, std::min(struct_size, sizeof(afis_timing)));
    return AFIS_OK;
}

int afis_get_option(const afis_ctx* ctx, const char* name, int64_t* value)
{
    if (!ctx || !name || !value) return AFIS_EINVAL;
    const std::string n(name);
    if (n == "adc_variant") *value = ctx->adc_variant;
    else if (n == "bound_cus") *value = (ctx->stream_lo && !ctx->overlap_failed) ? ctx->bound_cus : 0;
    else if (n == "search_timeout_s") *value = (int64_t)ctx->search_timeout_s;
    else if (n == "mf_blocks") *value = ctx->mf_blocks;
    else if (n == "query_batch") *value = ctx->query_batch;
    else if (n == "chunk") *value = ctx->chunk;
    else if (n == "tile_share") *value = ctx->tile_share;
    else if (n == "minu_generic") *value = ctx->minu_generic;
    else if (n == "minu_fast_max_latent") *value = rt_class_max_latent(4);            // read-only: what the fast candidate kernel's largest shape class takes (afis_device.h: rt_max_rows)
    else if (n == "minu_fast_max_rolled") *value = rt_class_max_rolled(4);
    else if (n == "minu_fast_max_cells") *value = 8192 * 4;
    else if (n == "mf_stats") *value = ctx->mf_collect_stats;
    else if (n == "rowmax_budget_mb") *value = ctx->rowmax_budget_bytes >> 20;
    else if (n == "lut_dtype") *value = 32;
    else return AFIS_EINVAL;
    return AFIS_OK;
}

int afis_set_option(afis_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return AFIS_EINVAL;
    const std::string n(name);
    if (n == "adc_variant") { if (value < 0 || value > 9 || value == 4 || value == 5) return fail(ctx, AFIS_EINVAL, "adc_variant must be 0..3, 6, 7, 8 or 9"); ctx->adc_variant = (int)value; }
    else if (n == "lut_dtype") { if (value != 32) return fail(ctx, AFIS_EINVAL, "lut_dtype: only 32 (exact) exists; the 16-bit tolerance path of rounds 1-2 missed its stated tolerance and was removed (the reduced-precision pass of BASELINE.json configs[4] is adc_variant 9 / 8: a bound, followed by exact values)"); }
    else if (n == "tile_share") { if (value < 0 || value > 32) return fail(ctx, AFIS_EINVAL, "tile_share must be 0 (auto) or 1..32"); ctx->tile_share = (int)value; }
    else if (n == "query_batch") { if (value < 0 || value > 256) return fail(ctx, AFIS_EINVAL, "query_batch must be 0 (auto) or 1..256"); ctx->query_batch = (int)value; }
    else if (n == "chunk") { if (value < 0 || value > 65536) return fail(ctx, AFIS_EINVAL, "chunk must be 0 (auto) or 1..65536"); ctx->chunk = (int)value; }
    else if (n == "minu_generic") { ctx->minu_generic = value ? 1 : 0; }
    else if (n == "mf_stats") { ctx->mf_collect_stats = value ? 1 : 0; }
    else if (n == "search_timeout_s") { ctx->search_timeout_s = (double)value; }        // <= 0: unbounded hipStreamSynchronize
    else if (n == "bound_cus") {                                           // 0 = off; 32..224 in steps of 32: the bound pass on a stream confined to that many CUs (value / 8 of every XCD), the minutiae stage beside it on the others
        if (value < 0 || value > 224 || (value & 31)) return fail(ctx, AFIS_EINVAL, "bound_cus must be 0 (off), 32, 64, ... 224 (the runtime honours CU masks in steps of 32 CUs: 4 per XCD)");
        if (value > 0 && value + 32 > ctx->n_cus) return fail(ctx, AFIS_EINVAL, "bound_cus must leave at least 32 of the device's CUs to the other stream");
        if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, AFIS_EDEVICE, "bound_cus: hipSetDevice failed");
        for (hipStream_t* ps : {&ctx->stream_lo, &ctx->stream_hi}) if (*ps) { (void)hipStreamSynchronize(*ps); (void)hipStreamDestroy(*ps); *ps = nullptr; }
        ctx->bound_cus = (int)value;
        ctx->overlap_failed = false;
        if (value > 0) {
            uint32_t lo[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the runtime deals the mask's bits round-robin over the XCDs: the low N bits are N / 8 CUs of each
            const bool whole_xcds = getenv("AFIS_BOUND_WHOLE_XCDS") != nullptr;   // experiment: value / 32 WHOLE XCDs for the bound pass instead of value / 8 CUs of each — measured: the pass takes 287 ms per group on 4 whole XCDs against 235 on 16 CUs of each of the 8 (the power limit acts per XCD); the step falls back to the back-to-back time
            for (int b = 0; b < std::min(256, ctx->n_cus); ++b) ((whole_xcds ? (b & 7) < (int)value / 32 : b < (int)value) ? lo : hi)[b >> 5] |= 1u << (b & 31);
            hipError_t e = hipExtStreamCreateWithCUMask(&ctx->stream_lo, 8, lo);
            if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ctx->stream_hi, 8, hi);
            if (e != hipSuccess) {
                for (hipStream_t* ps : {&ctx->stream_lo, &ctx->stream_hi}) if (*ps) { (void)hipStreamDestroy(*ps); *ps = nullptr; }
                ctx->bound_cus = 0;
                return fail(ctx, AFIS_EDEVICE, std::string("bound_cus: hipExtStreamCreateWithCUMask: ") + hipGetErrorString(e));
            }
        }
    }
    else if (n == "mf_blocks") { if (value != 2 && value != 3 && value != 102) return fail(ctx, AFIS_EINVAL, "mf_blocks must be 2, 3 or 102 (the software-pipelined bound pass)"); ctx->mf_blocks = (int)value; }
    else if (n == "rowmax_budget_mb") { if (value < 1) return fail(ctx, AFIS_EINVAL, "rowmax_budget_mb must be positive"); ctx->rowmax_budget_bytes = value << 20; }
    else return fail(ctx, AFIS_EINVAL, "unknown option: " + n);
    return AFIS_OK;
}

#ifdef AFIS_PARITY_TAPS   // the parity taps exist only in libafis_hip_test.so (include/afis_matcher_taps.h)
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
                                             d_out.as<MinuCand>(), d_n.as<int32_t>(), stage, s));
        } else {
            slot = which - 1; cap = kTopMinu;
            const size_t per_wg = 2 * (((size_t)std::max(1, grp.max_nL) * std::max(1, ctx->max_nR) + 63) / 64 * 64) + 4096;
            HIPCHK(ctx, ctx->scratch.ensure(per_wg * 4 * 64));
            HIPCHK(ctx, ctx->cands.ensure(3 * (size_t)kTopMinu * sizeof(MinuCand))); HIPCHK(ctx, ctx->cand_n.ensure(12)); HIPCHK(ctx, ctx->minu_fb.ensure(minu_fb_ints(3, 1) * 4));
            HIPCHK(ctx, launch_minu_cands(d, one, ctx->scratch.as<float>(), per_wg, 64, ctx->minu_generic, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->minu_fb.as<int32_t>(), grp.max_nL, ctx->max_nR, nullptr, s));
            HIPCHK(ctx, launch_graph_minutiae(d, one, ctx->cands.as<MinuCand>(), ctx->cand_n.as<int32_t>(), ctx->parts.as<float>(), nullptr, nullptr,
                                              d_out.as<MinuCand>(), d_n.as<int32_t>(), stage, s));
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

#endif  // AFIS_PARITY_TAPS

}  // extern "C"
