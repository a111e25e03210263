// afis_api.cpp — context life cycle, device identity, timing and options of the C ABI in include/afis_matcher.h.  The gallery side is afis_gallery.cpp, the search
// side afis_search.cpp, the parity taps (test library only) afis_taps.cpp; device code lives in adc*.hip, minu.hip, graph.hip and pq_encode.hip.  Host C++ only.
#include "afis_ctx.h"
#include <algorithm>
#include <vector>

using namespace afis;

namespace afis {
thread_local std::string g_create_error;
}

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
    register_context(c);
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
    unregister_context(c);
    (void)hipSetDevice(c->device);
    static const bool trace = getenv("AFIS_DESTROY_TRACE") != nullptr;       // development aid: where a destroy spends its time, on stderr
    auto lap = [&](const char* what) { if (trace) { fprintf(stderr, "afis_destroy: %s\n", what); fflush(stderr); } };
    // the side streams first, the context's stream last: work on the context's stream may wait for events of the side streams, and a blocking wait on it alone has been seen
    // not to return while they had not been waited for themselves (tools/repro/README.md)
    lap("waiting for the side streams");
    for (hipStream_t* ps : {&c->stream_lo, &c->stream_hi}) if (*ps) (void)hipStreamSynchronize(*ps);
    lap("waiting for the context's stream");
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    lap("destroying the side streams");
    for (hipStream_t* ps : {&c->stream_lo, &c->stream_hi}) if (*ps) { (void)hipStreamDestroy(*ps); *ps = nullptr; }
    lap("freeing device memory");
    for (afis_queries* q : c->parked_queries) { for (QueryGroup& g : q->groups) g.release(); delete q; }     // query groups of a search that timed out (the streams have been waited for above: this destroy may block where that search did)
    c->parked_queries.clear();
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_bytes = 0; }
    free_gallery_dev(c);
    c->codewords.release(); c->table.release(); c->lut.release(); c->rm_val.release(); c->rm_arg.release(); c->rm_cv.release(); c->rm_n.release();
    c->parts.release(); c->scores.release(); c->scratch.release(); c->cands.release(); c->cand_n.release(); c->minu_fb.release(); c->diag.release(); c->topk_idx.release(); c->topk_score.release(); c->lutq.release(); c->lutq_min.release(); c->lutq_rng.release(); c->lutq_rowc.release(); c->lut32.release();
    c->mf_cw16.release(); c->mf_cwn.release(); c->mf_bfrag.release(); c->mf_rowk.release(); c->mf_rec.release(); c->mf_stats.release();
    lap("destroying events and the context's stream");
    for (auto& e : c->evpool) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    lap("joining the staging thread");
    if (c->staging_reaper.joinable()) c->staging_reaper.join();
    lap("done");
    delete c;
}

const char* afis_last_error(const afis_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

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

int afis_rank_list(const float* scores, int64_t n, int ref_order, int k, int64_t* idx, float* sc)
{
    if (n < 0 || k < 0 || (n > 0 && !scores) || (k > 0 && !idx)) return fail(nullptr, AFIS_EINVAL, "afis_rank_list: null argument or negative size");
    if (n > 0x7fffffff) return fail(nullptr, AFIS_EINVAL, "afis_rank_list: more than 2^31 - 1 scores");
    std::vector<int> ind((size_t)n);
    for (int64_t i = 0; i < n; ++i) ind[(size_t)i] = (int)i;
    auto by_score = [scores](const int& a, const int& b) { return scores[a] > scores[b]; };     // matcher.cpp:306-308
    if (ref_order) std::sort(ind.begin(), ind.end(), by_score);
    else std::stable_sort(ind.begin(), ind.end(), by_score);
    for (int j = 0; j < k; ++j) {
        const bool in = j < n;
        idx[j] = in ? ind[(size_t)j] : -1;
        if (sc) sc[j] = in ? scores[ind[(size_t)j]] : 0.0f;
    }
    return AFIS_OK;
}


int afis_get_option(const afis_ctx* ctx, const char* name, int64_t* value)
{
    if (!ctx || !name || !value) return AFIS_EINVAL;
    const std::string n(name);
    if (n == "adc_variant") *value = ctx->adc_variant;
    else if (n == "bound_cus") *value = (ctx->stream_lo && !ctx->overlap_failed) ? ctx->bound_cus : 0;
    else if (n == "search_timeout_s") *value = ctx->search_timeout_s <= 0 ? 0 : (int64_t)std::ceil(ctx->search_timeout_s);     // rounded up: a bound set in milliseconds must not read back as 0 = "unbounded"
    else if (n == "search_timeout_ms") *value = ctx->search_timeout_s <= 0 ? 0 : (int64_t)std::llround(ctx->search_timeout_s * 1e3);
    else if (n == "mf_blocks") *value = ctx->mf_blocks;
    else if (n == "query_batch") *value = ctx->query_batch;
    else if (n == "chunk") *value = ctx->chunk;
    else if (n == "tile_share") *value = ctx->tile_share;
    else if (n == "minu_generic") *value = ctx->minu_generic;
    else if (n == "s3_tie_order") *value = ctx->s3_tie_order;
    else if (n == "ref_tie_order") *value = ctx->s3_tie_order + ctx->s89_tie_order;
    else if (n == "minu_fast_max_latent") *value = rt_class_max_latent(4);            // read-only: what the fast candidate kernel's largest shape class takes (afis_device.h: rt_max_rows)
    else if (n == "minu_fast_max_rolled") *value = rt_class_max_rolled(4);
    else if (n == "minu_fast_max_cells") *value = rt_class_simi_floats(4);
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
    if (n == "adc_variant") {
#ifdef AFIS_EXPERIMENTAL_KERNELS
        if (value < 0 || value > 9 || value == 4 || value == 5) return fail(ctx, AFIS_EINVAL, "adc_variant must be 0..3, 6, 7, 8 or 9");
#else
        if (value != 8 && value != 9) return fail(ctx, AFIS_EINVAL, "adc_variant must be 9 (matrix-core bound pass + exact values) or 8 (16-bit LDS-table bound pass + exact values); the direct kernels 0..3, 6, 7 "
                                                                     "are reference kernels built into libafis_hip_test.so only");
#endif
        ctx->adc_variant = (int)value;
    }
    else if (n == "lut_dtype") { if (value != 32) return fail(ctx, AFIS_EINVAL, "lut_dtype: only 32 (exact) exists; the 16-bit tolerance path of rounds 1-2 missed its stated tolerance and was removed (the reduced-precision pass of BASELINE.json configs[4] is adc_variant 9 / 8: a bound, followed by exact values)"); }
    else if (n == "tile_share") { if (value < 0 || value > 32) return fail(ctx, AFIS_EINVAL, "tile_share must be 0 (auto) or 1..32"); ctx->tile_share = (int)value; }
    else if (n == "query_batch") { if (value < 0 || value > 256) return fail(ctx, AFIS_EINVAL, "query_batch must be 0 (auto) or 1..256"); ctx->query_batch = (int)value; }
    else if (n == "chunk") { if (value < 0 || value > 65536) return fail(ctx, AFIS_EINVAL, "chunk must be 0 (auto) or 1..65536"); ctx->chunk = (int)value; }
    else if (n == "minu_generic") { ctx->minu_generic = value ? 1 : 0; }
    else if (n == "s3_tie_order") { if (value != 0 && value != 1) return fail(ctx, AFIS_EINVAL, "s3_tie_order must be 0 (equal candidate norms by ascending element index) or 1 (in the order libstdc++'s std::sort leaves them)"); ctx->s3_tie_order = (int)value; ctx->s89_tie_order = 0; }
    else if (n == "ref_tie_order") { if (value < 0 || value > 2) return fail(ctx, AFIS_EINVAL, "ref_tie_order must be 0 (equal sort keys by ascending index), 1 (candidate norms in the order libstdc++'s std::sort leaves them: = s3_tie_order 1) or 2 (the scores of the greedy selections of S8 and S9 as well)"); ctx->s3_tie_order = value >= 1; ctx->s89_tie_order = value >= 2; }
    else if (n == "mf_stats") { ctx->mf_collect_stats = value ? 1 : 0; }
    else if (n == "search_timeout_s") { ctx->search_timeout_s = (double)value; }        // <= 0: unbounded hipStreamSynchronize
    else if (n == "search_timeout_ms") { ctx->search_timeout_s = (double)value * 1e-3; }
    else if (n == "bound_cus") {                                           // 0 = off; 32..224 in steps of 32: the bound pass on a stream confined to that many CUs (value / 8 of every XCD), the minutiae stage beside it on the others
        if (value < 0 || value > 224 || (value & 31)) return fail(ctx, AFIS_EINVAL, "bound_cus must be 0 (off), 32, 64, ... 224 (the runtime honours CU masks in steps of 32 CUs: 4 per XCD)");
        if (value > 0 && value + 32 > ctx->n_cus) return fail(ctx, AFIS_EINVAL, "bound_cus must leave at least 32 of the device's CUs to the other stream");
        if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, AFIS_EDEVICE, "bound_cus: hipSetDevice failed");
        { const int rcd = drain_abandoned(ctx); if (rcd != AFIS_OK) return rcd; }            // a search that timed out may still run on the side streams: bounded wait, not the blocking one below
        for (hipStream_t* ps : {&ctx->stream_lo, &ctx->stream_hi}) if (*ps) { (void)hipStreamSynchronize(*ps); (void)hipStreamDestroy(*ps); *ps = nullptr; }
        ctx->bound_cus = (int)value;
        ctx->overlap_failed = false;
        if (value > 0) {
            uint32_t lo[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the runtime deals the mask's bits round-robin over the XCDs: the low N bits are N / 8 CUs of each
            const bool whole_xcds = AFIS_EXPERIMENT_ENV("AFIS_BOUND_WHOLE_XCDS") != nullptr;   // experiment: value / 32 WHOLE XCDs for the bound pass instead of value / 8 CUs of each — measured: the pass takes 287 ms per group on 4 whole XCDs against 235 on 16 CUs of each of the 8 (the power limit acts per XCD); the step falls back to the back-to-back time
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
    else if (n == "mf_blocks") {
#ifdef AFIS_EXPERIMENTAL_KERNELS
        if (value != 2 && value != 3) return fail(ctx, AFIS_EINVAL, "mf_blocks must be 2 or 3 (row blocks per wave of the bound pass)");
#else
        if (value != 2) return fail(ctx, AFIS_EINVAL, "mf_blocks must be 2 (the three-row-block form of the bound pass is built into libafis_hip_test.so only; it measured 1 % faster alone and 2 % slower in the default schedule)");
#endif
        ctx->mf_blocks = (int)value;
    }
    else if (n == "rowmax_budget_mb") { if (value < 1) return fail(ctx, AFIS_EINVAL, "rowmax_budget_mb must be positive"); ctx->rowmax_budget_bytes = value << 20; }
    else return fail(ctx, AFIS_EINVAL, "unknown option: " + n);
    return AFIS_OK;
}

}  // extern "C"
