// afis_ctx.h — the host side's internal state and helpers, shared by its translation units: afis_api.cpp (context, options, timing), afis_gallery.cpp (staging,
// container, commit), afis_search.cpp (query groups, the launch sequence of a search, correspondences, all-templates mode) and afis_taps.cpp (parity taps, test library only).
// Not part of the ABI: include/afis_matcher.h is.
#pragma once
#include "../../include/afis_matcher.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include <sys/mman.h>

#include "afis_device.h"
#include "template_io.h"
#include <memory>

namespace afis {

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

}  // namespace afis

using namespace afis;       // (an internal header: every host unit that includes it works inside this namespace)

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
    void* h_pin = nullptr; size_t h_pin_bytes = 0;   // pinned host buffer for what a search reads back inside its wait (rank lists, diagnostics)
    std::vector<float> h_scores, h_parts;
    int adc_variant = 9;                 // 9: fp16 matrix-core bound pass + exact recomputation (default); 8: 16-bit LDS-table bound pass + exact refine; 7: direct exact kernel; 0-3, 6: earlier direct kernels
    int tile_share = 0;                  // adc_variant 8: consecutive chunks per tile on an XCD; 0 = 4 (the refine's fp32 table stays in L2)
    int query_batch = 0;                 // latents per launch group at most; 0 = by shard size (afis_queries_upload); adc_variant 9 places the cuts by latent texture rows
    int chunk = 0;                       // gallery templates per ADC workgroup; 0 = by gallery size
    int minu_generic = 0;
    int s3_tie_order = 0;                // 1: equal candidate norms in the order libstdc++'s std::sort leaves them (matcher.cpp:476); 0: ascending element index
    int s89_tie_order = 0;               // 1 (option ref_tie_order 2): the greedy selections of S8 and S9 walk equal scores in std::sort's order too (graph.hip::sort_scores)
    double search_timeout_s = 600.0;     // bound on every host wait of a search (AFIS_SEARCH_TIMEOUT_S; <= 0: plain hipStreamSynchronize, unbounded)
    int64_t planned_group_bytes = 0;     // per-group buffers the largest launch group uploaded so far will take (group_budget_bytes of OTHER contexts on the device leaves room for it)
    std::vector<afis_queries*> parked_queries;   // query groups a timed-out search may still be reading: freed by drain_abandoned() once the device is back (afis_queries_free parks them here)
    bool search_abandoned = false;       // the last search returned at its deadline: the device may still be working on it (the next search waits for it first)
    bool overlap_failed = false;         // a wait of the overlapped schedule timed out: later searches keep to one stream
    double overlap_cell_ratio = 0.037;   // a launch group runs the overlapped schedule while (latent x rolled minutiae cells) <= this x (latent texture rows x rolled texture points); AFIS_OVERLAP_CELL_RATIO
    int64_t rowmax_budget_bytes = 0;     // device memory a launch group's per-pair buffers may take (option rowmax_budget_mb); 0 = 60 % of what hipMemGetInfo reports free
    afis_timing timing = {};
};

namespace afis {

extern thread_local std::string g_create_error;   // last afis_create failure of THIS thread (there is no context to hang it on)

inline int fail(afis_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}
#define HIPCHK(ctx, call)                                                                                       \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                          \
        return fail(ctx, AFIS_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e_) + (e_ == hipErrorOutOfMemory ?                                      \
            " (device memory: a launch group's buffers are sized from the memory that was free when the queries were uploaded - with several contexts or processes on one "  \
            "device lower option rowmax_budget_mb or query_batch)" : "")); } while (0)

template <class T, class A>
inline hipError_t upload(DevBuf& b, const std::vector<T, A>& v, hipStream_t s)
{
    hipError_t e = b.ensure(std::max<size_t>(v.size() * sizeof(T), 16));
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
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

constexpr int64_t kMfRecBytesPerRow = 8;               // adc_variant 9: one 8-byte record per (rolled template, latent texture row)
// afis_gallery.cpp
int materialise(afis_ctx* ctx);                        // the staged gallery as host arrays: a container that afis_gallery_load only mapped is copied into ctx->hg now
void free_gallery_dev(afis_ctx* c);
int ensure_mf_gallery(afis_ctx* ctx, hipStream_t s);   // adc_variant 9's tile-aligned copy of the gallery codes (built at commit, or by the first search after the variant was selected)
void views_of(const HostTemplate& t, std::vector<afis_minutiae_view>& mv, std::vector<afis_texture_view>& tv, afis_template_view& out);
// afis_search.cpp
// spec == NULL: the reference's selection for every query (templates 27, 3, 12 and texture template 0, matcher.cpp:380-415).
// spec != NULL (afis_match_all_templates): query i uses latent minutiae templates spec[i*4 + 0..2] (-1 = none) and latent texture
// template spec[i*4 + 3] (-1 = none), and is never "latent empty".
int build_group(afis_ctx* ctx, const afis_template_view* qs, int nq, QueryGroup& grp, std::vector<int32_t>& status_out, const int* spec = nullptr);
int adc_stage_q(afis_ctx* ctx, QueryGroup& grp, int chunk, bool exact, hipEvent_t after_lut = nullptr);
int adc_refine_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, bool compact);
int adc_stage_mfma(afis_ctx* ctx, QueryGroup& grp, bool all_rows, hipEvent_t after_lut = nullptr, hipEvent_t after_bound = nullptr, bool compact = false, hipStream_t sb = nullptr,
                   bool refine_now = true, unsigned long long* diag = nullptr);
#ifdef AFIS_EXPERIMENTAL_KERNELS
int ensure_codes_cf(afis_ctx* ctx, int variant);       // the direct kernels' lane-ordered code stream (adc_variant 6 / 7), laid out at first use
#endif
int wait_streams(afis_ctx* ctx, std::initializer_list<hipStream_t> streams, const char* what);
void register_context(afis_ctx* ctx); void unregister_context(afis_ctx* ctx);   // the process-wide list group_budget_bytes consults
int drain_abandoned(afis_ctx* ctx);                     // waits (bounded) for a search that returned at its deadline

}  // namespace afis
