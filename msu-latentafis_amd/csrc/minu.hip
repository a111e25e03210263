// minu.hip — candidate generation of the minutiae-template scorer and the final score fusion:
//   S1  descriptor similarity  simi = max(0, A * B^T)              (matching/matcher.cpp:440-452)
//   S2  normalisation          norm = s / (rowsum + colsum - s + 1e-6) (matcher.cpp:455-470)
//   S3  the 120 correspondences with the largest norm, in rank order (matcher.cpp:473-488)
//   S10 fusion                 final = score[0] + score[1] + score[2] + score[28]*0.3   (matcher.cpp:376-417, :188)
// The correspondence lists go to HBM (120 x 8 B per task) and are consumed by k_graph_minutiae (graph.hip).
//
// One 256-thread workgroup per (query, selected latent template, gallery template) task, persistent over a strided task list.
// Canonical arithmetic (see oracle/afis_oracle.cpp): descriptor dot products are k-ascending fmaf chains, row/column sums are
// index-ascending, the normalisation is evaluated in double exactly as the reference's expression promotes it.
#include "afis_device.h"

namespace afis {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
typedef unsigned long long u64;

#ifdef AFIS_PHASE_TIMING
__device__ u64 g_phase_cycles[32];
#define PHASE_INIT() u64 ph_t0 = __builtin_readcyclecounter()
#define PHASE(i) do { if (threadIdx.x == 0) { const u64 ph_t1 = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], ph_t1 - ph_t0); ph_t0 = ph_t1; } } while (0)
#else
#define PHASE_INIT() do {} while (0)
#define PHASE(i) do {} while (0)
#endif

__device__ __forceinline__ uint32_t ord_f32(float v)
{
    v = v + 0.0f;                                   // -0 -> +0 so that equal floats get equal keys
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Workgroup-wide sum of wave-uniform partial counts (each wave passes its own total).  One barrier per call.
__device__ __forceinline__ int wg_sum(int wave_total, int* s_slots /*[2][kWaves]*/, int& parity)
{
    if ((threadIdx.x & 63) == 0) s_slots[parity * kWaves + (threadIdx.x >> 6)] = wave_total;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) tot += s_slots[parity * kWaves + w];
    parity ^= 1;
    return tot;
}
__device__ __forceinline__ int wave_popc(bool p) { return __popcll(__ballot(p)); }

constexpr int kGemmRows = 64;       // latent rows per GEMM tile
constexpr int kGemmCols = 32;       // rolled columns per GEMM tile
constexpr int kGemmLd = 100;        // padded row stride (floats): 16-byte aligned rows, conflict-free b128 column walks
constexpr int kKeyRegs = 16;        // per-thread keys held in registers when nL*nR <= 256*16
constexpr int kFastL = 64, kFastR = 128;          // pair shapes whose whole similarity matrix stays in LDS
constexpr int kFastN = kFastL * kFastR;

struct MinuSmem {
    float A[kGemmRows * kGemmLd];   // 25.6 KB
    float B[kGemmCols * kGemmLd];   // 12.8 KB
    float simi[kFastN];             // 32 KB
    float rowsum[kFastL];
    float colsum[kFastR];
    u64 keys[128];
    int te[kTopMinu];
    int slots[2 * kWaves];
    int counter;
};

// Global scratch of one workgroup (pairs too large for the LDS fast path): simi[n] | keys[n] | rowsum[2048] | colsum[2048]
__global__ __launch_bounds__(kThreads) void k_minu_cands(QueryDev q, GalleryDev g, float* __restrict__ scratch, size_t scratch_per_wg,
                                                         MinuCand* __restrict__ cands, int32_t* __restrict__ cand_n, int skip_fast)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MinuSmem& sm = *reinterpret_cast<MinuSmem*>(smem_raw);
    float* gscr = scratch + (size_t)blockIdx.x * scratch_per_wg;
    int parity = 0;
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    const int tid = threadIdx.x;
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        // task order: gallery template fastest, then selected template, then query
        const int gi = (int)(task % g.G);
        const int qs = (int)(task / g.G);                    // qi*3 + s
        const int l0 = q.lm_off[qs], nL = q.lm_off[qs + 1] - l0;
        const int r0 = g.minu_off[gi], nR = g.minu_off[gi + 1] - r0;
        if (nL <= 0 || nR <= 0) { if (tid == 0) cand_n[task] = 0; continue; }     // matcher.cpp:400-404
        if (skip_fast && nL <= kFastL && nR <= kFastR) continue;                   // done by k_minu_cands_fast
        const int n = nL * nR;
        PHASE_INIT();
        const bool fast = nL <= kFastL && nR <= kFastR;
        const size_t half = (scratch_per_wg - 4096) >> 1;
        float* simi = fast ? sm.simi : gscr;
        uint32_t* gkeys = reinterpret_cast<uint32_t*>(gscr + half);
        float* rowsum = fast ? sm.rowsum : gscr + 2 * half;
        float* colsum = fast ? sm.colsum : gscr + 2 * half + 2048;

        // ---- S1: simi = max(0, A * B^T), canonical order = fmaf chain, k ascending (matcher.cpp:440-452) ----
        for (int it = 0; it < nL; it += kGemmRows) {
            __syncthreads();
            for (int e = tid; e < kGemmRows * (kDes / 4); e += kThreads) {
                const int r = e / (kDes / 4), k4 = e - r * (kDes / 4);
                float4 a = make_float4(0, 0, 0, 0);
                if (it + r < nL) a = *reinterpret_cast<const float4*>(q.lm_des + (size_t)(l0 + it + r) * kDes + k4 * 4);
                *reinterpret_cast<float4*>(&sm.A[r * kGemmLd + k4 * 4]) = a;
            }
            for (int jt = 0; jt < nR; jt += kGemmCols) {
                if (jt) __syncthreads();
                for (int e = tid; e < kGemmCols * (kDes / 4); e += kThreads) {
                    const int r = e / (kDes / 4), k4 = e - r * (kDes / 4);
                    float4 b = make_float4(0, 0, 0, 0);
                    if (jt + r < nR) b = *reinterpret_cast<const float4*>(g.minu_des + (size_t)(r0 + jt + r) * kDes + k4 * 4);
                    *reinterpret_cast<float4*>(&sm.B[r * kGemmLd + k4 * 4]) = b;
                }
                __syncthreads();
                const int ty = tid >> 3, tx = tid & 7;               // rows ty + 32*r (r < 2), cols tx + 8*c (c < 4)
                float acc[2][4];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
#pragma unroll 4
                for (int k4 = 0; k4 < kDes / 4; ++k4) {
                    float4 a[2], b[4];
#pragma unroll
                    for (int r = 0; r < 2; ++r) a[r] = *reinterpret_cast<const float4*>(&sm.A[(ty + 32 * r) * kGemmLd + k4 * 4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(&sm.B[(tx + 8 * c) * kGemmLd + k4 * 4]);
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = acc[r][c];
                            v = fmaf(a[r].x, b[c].x, v); v = fmaf(a[r].y, b[c].y, v);
                            v = fmaf(a[r].z, b[c].z, v); v = fmaf(a[r].w, b[c].w, v);
                            acc[r][c] = v;
                        }
                }
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = it + ty + 32 * r, j = jt + tx + 8 * c;
                        if (i < nL && j < nR) { float v = acc[r][c]; if (v < 0) v = 0; simi[(size_t)i * nR + j] = v; }
                    }
            }
        }
        __syncthreads();
        PHASE(0);
        // ---- S2: column sums (rolled) / row sums (latent), index ascending (:455-456) ----
        for (int j = tid; j < nR; j += kThreads) {
            float sacc = 0.f;
#pragma unroll 8
            for (int i = 0; i < nL; ++i) sacc += simi[(size_t)i * nR + j];
            colsum[j] = sacc;
        }
        for (int i = tid; i < nL; i += kThreads) {
            float sacc = 0.f;
#pragma unroll 8
            for (int j = 0; j < nR; ++j) sacc += simi[(size_t)i * nR + j];
            rowsum[i] = sacc;
        }
        __syncthreads();
        PHASE(1);
        // ---- S3: top-120 by normalised similarity (:461-488) ----
        const int topN = n < kTopMinu ? n : kTopMinu;
        auto norm_key = [&](int e) {
            const int i = e / nR, j = e - i * nR;
            const float sv = simi[e];
            float f = rowsum[i] + colsum[j];
            f = f - sv;
            return ord_f32((float)((double)sv / ((double)f + 0.000001)));                        // :467
        };
        // K-th largest key T, built bit by bit from wave-level ballot counts; then the keys above T plus the lowest-index
        // keys equal to T.  `keyv(u)` is this thread's u-th key (element e = tid + u*256), 0 beyond n (real keys are >= 2^31).
        uint32_t rk[kKeyRegs];
        const bool in_regs = n <= kThreads * kKeyRegs;
        const int n_u = in_regs ? kKeyRegs : (n + kThreads - 1) / kThreads;
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < kKeyRegs; ++u) { const int e = tid + u * kThreads; rk[u] = e < n ? norm_key(e) : 0u; }
        } else {
            for (int e = tid; e < n; e += kThreads) gkeys[e] = norm_key(e);
            __syncthreads();
        }
        PHASE(2);
        auto count_if = [&](auto pred) {                    // pred(key, e) -> bool; returns the workgroup-wide count
            int c = 0;
            if (in_regs) {
#pragma unroll
                for (int u = 0; u < kKeyRegs; ++u) c += wave_popc(pred(rk[u], (uint32_t)(tid + u * kThreads)));
            } else {
                for (int u = 0; u < n_u; ++u) { const int e = tid + u * kThreads; c += wave_popc(e < n && pred(gkeys[e < n ? e : 0], (uint32_t)e)); }
            }
            return wg_sum(c, sm.slots, parity);
        };
        uint32_t T = 0x80000000u;                           // every real key has the top bit set (norm >= 0)
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            if (count_if([cand](uint32_t k, uint32_t) { return k >= cand; }) >= topN) T = cand;
        }
        const int n_gt = count_if([T](uint32_t k, uint32_t) { return k > T; });
        const int n_eq = count_if([T](uint32_t k, uint32_t) { return k == T; });
        const int need = topN - n_gt;                       // >= 1
        uint32_t Bnd = 0xffffffffu;                         // keep the keys equal to T whose index is <= Bnd
        if (n_eq != need) {
            Bnd = 0;
            for (int bit = 30; bit >= 0; --bit) {           // Bnd = largest bound with count(key == T && e < Bnd) < need
                const uint32_t cand = Bnd | (1u << bit);
                if (count_if([T, cand](uint32_t k, uint32_t e) { return k == T && e < cand; }) < need) Bnd = cand;
            }
        }
        if (tid == 0) sm.counter = 0;
        __syncthreads();
        auto emit = [&](uint32_t k, int e) {
            if (e < n && (k > T || (k == T && (uint32_t)e <= Bnd))) {
                const int pos = atomicAdd(&sm.counter, 1);
                sm.keys[pos] = ((u64)k << 32) | (uint32_t)(~(uint32_t)e);
                sm.te[pos] = e;
            }
        };
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < kKeyRegs; ++u) emit(rk[u], tid + u * kThreads);
        } else {
            for (int e = tid; e < n; e += kThreads) emit(gkeys[e], e);
        }
        __syncthreads();
        PHASE(3);
        if (tid < topN) {                                   // rank by counting: the list leaves in rank order
            const u64 mine = sm.keys[tid];
            int r = 0;
#pragma unroll 8
            for (int k = 0; k < topN; ++k) r += sm.keys[k] > mine;
            const int e = sm.te[tid];
            const int i1 = e / nR, i2 = e - i1 * nR;
            MinuCand c; c.sim = simi[e]; c.li = (short)i1; c.ri = (short)i2;
            cands[(size_t)task * kTopMinu + r] = c;
        }
        if (tid == 0) cand_n[task] = topN;
        __syncthreads();
        PHASE(4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast path: pairs with nL <= 64 latent and nR <= 128 rolled minutiae (every template the extraction normally produces).
//   S1  the one dense contraction of the path (the reference's Eigen GEMM) runs on the matrix cores with the exact-fp32
//       v_mfma_f32_16x16x4_f32: wave w owns the 16-column tiles w, w+4, ...; fragments come straight from HBM/L2 (k-permuted
//       descriptor copies), results go to the LDS-resident similarity matrix.  No LDS staging, no barriers.
//   S2  sums and S3 keys from the LDS-resident similarity matrix.
//   S3  top-120 in two stages with ONE barrier: every wave finds the 120 largest of its own quarter of the keys (bit-by-bit
//       threshold search on ballot/popcount counts, keys in registers), wave 0 then takes the 120 largest of those <= 480 and
//       ranks them.  Keys are 45-bit composites (norm key << 13 | 8191 - element index): unique, so "larger = earlier" is
//       exactly "norm descending, index ascending" and no tie handling is needed.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFastU = kFastN / kThreads;            // 32 keys per lane at most
constexpr int kStage1Cap = 160;                      // stage-1 survivors per wave: a superset of the wave's 120 largest keys
constexpr int kHistBins = 2048;                      // histogram path: key bits 29..19 (7 exponent + 4 mantissa bits; norm < 1)
constexpr int kBndCap = 128;                         // keys that may share the threshold bin before the two-stage path takes over
constexpr int kHistMinN = 512;                       // below this the two-stage path is as cheap
struct FastSmem {
    float simi[kFastN];                               // 32 KB
    float rowsum[kFastL];
    float colsum[kFastR];
    union {
        u64 list[kWaves * kStage1Cap];                // two-stage path: stage-1 survivors
        struct { uint32_t hist[kHistBins / 2]; u64 bnd[kBndCap]; } h;   // histogram path: 2048 16-bit bins; keys of the threshold bin
    } sel;
    u64 top[128];                                     // the selected keys (unordered on the histogram path)
    int counts[kWaves];
    int wave_tot[kWaves];
    int thr_bin, c_above, n_top, n_bnd;
};

// K-th largest of the wave's composite keys c[0..U) (0 = padding), K >= 1 and K <= number of non-zero keys
template <int U>
__device__ __forceinline__ u64 wave_kth_largest(const u64 (&c)[U], int K)
{
    u64 T = 0;
    for (int bit = 44; bit >= 0; --bit) {
        const u64 cand = T | (1ull << bit);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) cnt += wave_popc(c[u] >= cand);
        if (cnt >= K) T = cand;
    }
    return T;
}
__device__ __forceinline__ int lane_prefix(u64 mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0)); }

// Stage 1 of the top-120 selection for one wave: keys of the elements e = (u*4 + wave)*64 + lane, u < U, are computed into
// registers; the wave's Kw largest (ties: lowest element index) are written to list[] as 45-bit composites.  Returns Kw.
template <int U>
__device__ __forceinline__ int wave_stage1(const FastSmem& sm, u64* list, int n, int nR, int wave, int lane)
{
    uint32_t rk[U];                                                      // 32-bit norm keys; the element index is implied by (u, lane)
    int n_own = 0;
    {
        // (i, j) of the lane's first element and the step between consecutive elements (256), without per-element divisions
        int e = wave * 64 + lane;
        int i = e / nR, j = e - i * nR;
        const int si = kThreads / nR, sj = kThreads - si * nR;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t key = 0;                                            // real keys have the top bit set
            if (e < n) {
                const float sv = sm.simi[e];
                float f = sm.rowsum[i] + sm.colsum[j];
                f = f - sv;
                key = ord_f32((float)((double)sv / ((double)f + 0.000001)));                    // matcher.cpp:467
            }
            rk[u] = key;
            n_own += wave_popc(e < n);
            e += kThreads; i += si; j += sj; if (j >= nR) { j -= nR; ++i; }
        }
    }
    const int Kw = n_own < kTopMinu ? n_own : kTopMinu;
    int n_out = 0;
    if (Kw > 0) {
        // norm lies in [0, 1): every key is in [0x80000000, 0xBF800000), so bit 31 is set and bit 30 clear.
        // Stage 2 only needs a SUPERSET of the wave's Kw largest keys, so the bit-by-bit threshold search stops as soon as the
        // keys >= T fit the wave's list (kStage1Cap): typically after 10-14 of the 30 bits.
        uint32_t T = 0x80000000u;
        int c_ge = n_own;                                                // keys >= T (padding keys are 0 < T)
        for (int bit = 29; bit >= 0 && c_ge > kStage1Cap; --bit) {
            const uint32_t cand = T | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) cnt += wave_popc(rk[u] >= cand);
            if (cnt >= Kw) { T = cand; c_ge = cnt; }
        }
        int need = kStage1Cap;                                           // of the keys equal to T (all of them if everything fits)
        int n_gt = 0;
        if (c_ge > kStage1Cap) {                                         // T is exact and too many keys equal it: keep the lowest indices
#pragma unroll
            for (int u = 0; u < U; ++u) n_gt += wave_popc(rk[u] > T);
            need = Kw - n_gt;
            n_out = Kw;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) n_gt += wave_popc(rk[u] > T);
            n_out = c_ge;
        }
        int base_gt = 0, base_eq = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {                                    // (u, lane) ascending = element index ascending
            const int e = (u * kWaves + wave) * 64 + lane;
            const bool gt = rk[u] > T, eq = rk[u] == T;
            const u64 mg = __ballot(gt), me = __ballot(eq);
            int pos = -1;
            if (gt) pos = base_gt + lane_prefix(mg);
            else if (eq) { const int r = base_eq + lane_prefix(me); if (r < need) pos = n_gt + r; }
            if (pos >= 0) list[pos] = ((u64)rk[u] << 13) | (u64)(8191 - e);
            base_gt += __popcll(mg); base_eq += __popcll(me);
        }
    }
    return n_out;
}

// Top-120 selection by histogram (all four waves, no serial stage): a 2048-bin histogram of the norm keys' bits 29..19 in LDS,
// a suffix scan that finds the bin in which the 120th largest key lies, then every key in a higher bin is selected outright and
// the (few) keys of the threshold bin are ranked among themselves as 45-bit composites (norm key, then lowest element index), so
// the result is exactly the 120 largest in the reference's order.  Returns false — uniformly for the workgroup — when the
// threshold bin is the zero bin or holds more than kBndCap keys; the caller then runs the two-stage path.
template <int U>
__device__ __forceinline__ bool select_hist(FastSmem& sm, int n, int nR, int wave, int lane, int tid)
{
    uint32_t rk[U];
    {
        int e = wave * 64 + lane;
        int i = e / nR, j = e - i * nR;
        const int si = kThreads / nR, sj = kThreads - si * nR;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t key = 0;
            if (e < n) {
                const float sv = sm.simi[e];
                float f = sm.rowsum[i] + sm.colsum[j];
                f = f - sv;
                key = ord_f32((float)((double)sv / ((double)f + 0.000001)));                    // matcher.cpp:467
            }
            rk[u] = key;
            e += kThreads; i += si; j += sj; if (j >= nR) { j -= nR; ++i; }
        }
    }
    for (int w = tid; w < kHistBins / 2; w += kThreads) sm.sel.h.hist[w] = 0u;
    if (tid == 0) { sm.n_top = 0; sm.n_bnd = 0; sm.thr_bin = -1; sm.c_above = 0; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (u * kWaves + wave) * 64 + lane;
        if (e < n) { const uint32_t bin = (rk[u] >> 19) & (kHistBins - 1); atomicAdd(&sm.sel.h.hist[bin >> 1], 1u << ((bin & 1) * 16)); }
    }
    __syncthreads();
    // thread tid owns bins 8*tid .. 8*tid+7; higher thread = larger keys
    int cnt[8], own = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = sm.sel.h.hist[4 * tid + w];
        cnt[2 * w] = (int)(x & 0xffffu); cnt[2 * w + 1] = (int)(x >> 16);
        own += cnt[2 * w] + cnt[2 * w + 1];
    }
    int suf = own;                                                       // keys in the bins of lanes >= this one (within the wave)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_down(suf, off); if (lane + off < 64) suf += t; }
    if (lane == 0) sm.wave_tot[wave] = suf;
    __syncthreads();
    int above = suf - own;                                               // keys in all higher bins
#pragma unroll
    for (int w = 0; w < kWaves; ++w) if (w > wave) above += sm.wave_tot[w];
#pragma unroll
    for (int b = 7; b >= 0; --b) {
        if (above < kTopMinu && above + cnt[b] >= kTopMinu) { sm.thr_bin = 8 * tid + b; sm.c_above = above; }
        above += cnt[b];
    }
    __syncthreads();
    const int B = sm.thr_bin, c_above = sm.c_above;
    if (B <= 0) return false;                                            // the 120th key is a zero norm: index order decides, two-stage path
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (u * kWaves + wave) * 64 + lane;
        if (e < n) {
            const int bin = (int)((rk[u] >> 19) & (kHistBins - 1));
            const u64 comp = ((u64)rk[u] << 13) | (u64)(8191 - e);
            if (bin > B) sm.top[atomicAdd(&sm.n_top, 1)] = comp;
            else if (bin == B) { const int p = atomicAdd(&sm.n_bnd, 1); if (p < kBndCap) sm.sel.h.bnd[p] = comp; }
        }
    }
    __syncthreads();
    const int n_bnd = sm.n_bnd;
    if (n_bnd > kBndCap) return false;
    const int need = kTopMinu - c_above;                                 // 1 <= need <= n_bnd
    if (tid < n_bnd) {
        const u64 mine = sm.sel.h.bnd[tid];
        int r = 0;
        for (int k = 0; k < n_bnd; ++k) r += sm.sel.h.bnd[k] > mine;
        if (r < need) sm.top[c_above + r] = mine;
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(kThreads, 4) void k_minu_cands_fast(QueryDev q, GalleryDev g, const float* __restrict__ lat_desp,
                                                              const float* __restrict__ rol_desp,   // k-permuted descriptor copies
                                                              MinuCand* __restrict__ cands, int32_t* __restrict__ cand_n)
{
    __shared__ FastSmem sm;
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int gi = (int)(task % g.G);
        const int qs = (int)(task / g.G);
        const int l0 = q.lm_off[qs], nL = q.lm_off[qs + 1] - l0;
        const int r0 = g.minu_off[gi], nR = g.minu_off[gi + 1] - r0;
        if (nL <= 0 || nR <= 0) { if (tid == 0) cand_n[task] = 0; continue; }     // matcher.cpp:400-404
        if (nL > kFastL || nR > kFastR) continue;                                  // left to k_minu_cands
        const int n = nL * nR;
        PHASE_INIT();
        // ---- S1 (matcher.cpp:440-452): simi = max(0, A * B^T) on the matrix cores ----
        // v_mfma_f32_16x16x4_f32 is exact fp32: D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))), i.e. bit for bit the
        // k-ascending fmaf chain of the oracle (CDNA4 guide §3 "FP32-input MFMA").  Operand layout: lane l supplies A[i = l&15][k = l>>4]
        // and B[k = l>>4][j = l&15]; step s covers k = 4s .. 4s+3.  lat_desp / rol_desp hold every descriptor with its 96 values
        // permuted as [g][s] = des[4s + g], so the 24 values lane group g needs are contiguous (six 16-byte loads).
        {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int li = lane & 15, lg = lane >> 4;
            const int n_it = (nL + 15) >> 4, n_jt = (nR + 15) >> 4;
            // work item = (column tile jt, pair of row tiles): the two row tiles are independent accumulator chains (an MFMA needs
            // 40 cycles before its result can be accumulated into again), and items are dealt round-robin to the four waves
            const int n_ip = (n_it + 1) >> 1;
            for (int item = wave; item < n_jt * n_ip; item += kWaves) {
                const int jt = item / n_ip, it0 = (item - jt * n_ip) * 2;
                const bool two = it0 + 1 < n_it;
                const int jr = min(jt * 16 + li, nR - 1);
                const int ir0 = min(it0 * 16 + li, nL - 1), ir1 = min(it0 * 16 + 16 + li, nL - 1);
                const float4* bp = reinterpret_cast<const float4*>(rol_desp + (size_t)(r0 + jr) * kDes + lg * 24);
                const float4* ap0 = reinterpret_cast<const float4*>(lat_desp + (size_t)(l0 + ir0) * kDes + lg * 24);
                const float4* ap1 = reinterpret_cast<const float4*>(lat_desp + (size_t)(l0 + ir1) * kDes + lg * 24);
                float bf[24], af0[24], af1[24];
#pragma unroll
                for (int v = 0; v < 6; ++v) {
                    const float4 x = bp[v], y = ap0[v], z = ap1[v];
                    bf[4 * v] = x.x; bf[4 * v + 1] = x.y; bf[4 * v + 2] = x.z; bf[4 * v + 3] = x.w;
                    af0[4 * v] = y.x; af0[4 * v + 1] = y.y; af0[4 * v + 2] = y.z; af0[4 * v + 3] = y.w;
                    af1[4 * v] = z.x; af1[4 * v + 1] = z.y; af1[4 * v + 2] = z.z; af1[4 * v + 3] = z.w;
                }
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 24; ++st) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af0[st], bf[st], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af1[st], bf[st], acc1, 0, 0, 0);
                }
                const int j = jt * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {                          // D: col = lane & 15, row = (lane >> 4) * 4 + r
                    const int i0 = it0 * 16 + lg * 4 + r, i1 = i0 + 16;
                    float v0 = acc0[r], v1 = acc1[r];
                    if (v0 < 0) v0 = 0;
                    if (v1 < 0) v1 = 0;
                    if (i0 < nL && j < nR) sm.simi[i0 * nR + j] = v0;
                    if (two && i1 < nL && j < nR) sm.simi[i1 * nR + j] = v1;
                }
            }
        }
        __syncthreads();
        PHASE(16);
        // ---- S2 (:455-456): index-ascending sums ----
        if (tid < nR) {
            float sacc = 0.f;
#pragma unroll 8
            for (int i = 0; i < nL; ++i) sacc += sm.simi[i * nR + tid];
            sm.colsum[tid] = sacc;
        } else if (tid >= 128 && tid - 128 < nL) {
            const int i = tid - 128;
            float sacc = 0.f;
#pragma unroll 8
            for (int j = 0; j < nR; ++j) sacc += sm.simi[i * nR + j];
            sm.rowsum[i] = sacc;
        }
        __syncthreads();
        PHASE(17);
        // ---- S3 (:461-488): the 120 largest norm values, ties by lowest element index; element e = (u*4 + wave)*64 + lane ----
        const int topN = n < kTopMinu ? n : kTopMinu;
        bool selected = false;
        if (n >= kHistMinN) {                                                 // histogram path (all waves); falls through when it declines
            if (n <= 8 * kThreads) selected = select_hist<8>(sm, n, nR, wave, lane, tid);
            else if (n <= 16 * kThreads) selected = select_hist<16>(sm, n, nR, wave, lane, tid);
            else selected = select_hist<kFastU>(sm, n, nR, wave, lane, tid);
        }
        if (selected) {
            PHASE(18);
            // rank by counting, two threads per key: each counts the larger keys in one half of the list
            const int ki = tid >> 1, half = tid & 1;
            const u64 mine = ki < kTopMinu ? sm.top[ki] : 0ull;
            int r = 0;
            const int k0 = half * (kTopMinu / 2);
#pragma unroll 4
            for (int k = k0; k < k0 + kTopMinu / 2; ++k) r += sm.top[k] > mine;
            r += __shfl_xor(r, 1);
            if (half == 0 && ki < kTopMinu) {
                const int e = 8191 - (int)(mine & 8191);
                const int i1 = e / nR, i2 = e - i1 * nR;
                MinuCand cd; cd.sim = sm.simi[e]; cd.li = (short)i1; cd.ri = (short)i2;
                cands[(size_t)task * kTopMinu + r] = cd;
            }
            if (tid == 0) cand_n[task] = kTopMinu;
            __syncthreads();
            PHASE(20);
            continue;
        }
        // two-stage path: number of key slots per lane actually needed: 8 (n <= 2048), 16 (n <= 4096) or 32
        int Kw;
        if (n <= 8 * kThreads) Kw = wave_stage1<8>(sm, sm.sel.list + wave * kStage1Cap, n, nR, wave, lane);
        else if (n <= 16 * kThreads) Kw = wave_stage1<16>(sm, sm.sel.list + wave * kStage1Cap, n, nR, wave, lane);
        else Kw = wave_stage1<kFastU>(sm, sm.sel.list + wave * kStage1Cap, n, nR, wave, lane);
        PHASE(18);
        if (lane == 0) sm.counts[wave] = Kw;
        __syncthreads();
        PHASE(19);
        // ---- stage 2 (wave 0): the topN largest of the <= 640 survivors, then rank them ----
        if (wave == 0) {
            constexpr int V = kWaves * kStage1Cap / 64;                       // 10 keys per lane
            u64 d[V];
            int off[kWaves + 1]; off[0] = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) off[w + 1] = off[w] + sm.counts[w];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int p = v * 64 + lane;                                  // position in the concatenation of the four lists
                u64 key = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) if (p >= off[w] && p < off[w + 1]) key = sm.sel.list[w * kStage1Cap + p - off[w]];
                d[v] = key;
            }
            // topN largest composites = norm key descending, element index ascending.  Threshold search on the 32-bit keys;
            // among the keys equal to the threshold the lowest element indices win (second, 13-bit search, only when needed).
            uint32_t hk[V], he[V];
#pragma unroll
            for (int v = 0; v < V; ++v) { hk[v] = (uint32_t)(d[v] >> 13); he[v] = 8191u - (uint32_t)(d[v] & 8191); }
            uint32_t T = 0x80000000u;
            for (int bit = 29; bit >= 0; --bit) {
                const uint32_t cand = T | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int v = 0; v < V; ++v) cnt += wave_popc(hk[v] >= cand);
                if (cnt >= topN) T = cand;
            }
            int n_gt = 0, n_eq = 0;
#pragma unroll
            for (int v = 0; v < V; ++v) { n_gt += wave_popc(hk[v] > T); n_eq += wave_popc(hk[v] == T); }
            const int need = topN - n_gt;
            uint32_t Emax = 0xffffffffu;
            if (n_eq != need) {
                uint32_t X = 0;                                       // largest X with count(eq && e < X) < need
                for (int bit = 12; bit >= 0; --bit) {
                    const uint32_t cand = X | (1u << bit);
                    int cnt = 0;
#pragma unroll
                    for (int v = 0; v < V; ++v) cnt += wave_popc(hk[v] == T && he[v] < cand);
                    if (cnt < need) X = cand;
                }
                Emax = X;
            }
            int base = 0;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const bool take = hk[v] > T || (hk[v] == T && he[v] <= Emax);
                const u64 m = __ballot(take);
                if (take) sm.top[base + lane_prefix(m)] = d[v];
                base += __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // rank by counting; the list leaves in rank order
            u64 mine[2]; int r[2] = {0, 0};
            mine[0] = lane < topN ? sm.top[lane] : 0; mine[1] = lane + 64 < topN ? sm.top[lane + 64] : 0;
#pragma unroll 4
            for (int k = 0; k < topN; ++k) { const u64 kk = sm.top[k]; r[0] += kk > mine[0]; r[1] += kk > mine[1]; }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (lane + 64 * h < topN) {
                    const int e = 8191 - (int)(mine[h] & 8191);
                    const int i1 = e / nR, i2 = e - i1 * nR;
                    MinuCand cd; cd.sim = sm.simi[e]; cd.li = (short)i1; cd.ri = (short)i2;
                    cands[(size_t)task * kTopMinu + r[h]] = cd;
                }
            }
            if (lane == 0) cand_n[task] = topN;
        }
        __syncthreads();
        PHASE(20);
    }
}

hipError_t launch_minu_cands(const QueryDev& q, const GalleryDev& g, float* scratch, size_t scratch_floats_per_wg, int n_wg,
                             int max_nL, int max_nR, int force_generic, MinuCand* cands, int32_t* cand_n, hipStream_t stream)
{
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    if (n_tasks <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_minu_cands), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MinuSmem));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (!force_generic) {
        const int gridf = (int)(n_tasks < 8192 ? n_tasks : 8192);
        hipLaunchKernelGGL(k_minu_cands_fast, dim3(gridf), dim3(kThreads), 0, stream, q, g, q.lm_desp, g.minu_desp, cands, cand_n);
        if (max_nL <= kFastL && max_nR <= kFastR) return hipGetLastError();       // every pair took the fast path
    }
    const int grid = (int)(n_tasks < n_wg ? n_tasks : n_wg);
    hipLaunchKernelGGL(k_minu_cands, dim3(grid), dim3(kThreads), sizeof(MinuSmem), stream, q, g, scratch, scratch_floats_per_wg, cands, cand_n,
                       force_generic ? 0 : 1);
    return hipGetLastError();
}

// =====================================================================================================================
// S10 fusion: final = score[0] + score[1] + score[2] + score[28]*0.3 (matcher.cpp:188), where the reference's score
// vector holds the three minutiae scores at [0..2] and the texture score at index (#latent minutiae templates).
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_fuse(QueryDev q, GalleryDev g, const float* __restrict__ parts, float* __restrict__ scores)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)q.nq * g.G;
    if (idx >= n) return;
    const int qi = (int)(idx / g.G), gi = (int)(idx - (long long)qi * g.G);
    if (q.status[qi] != 0 || g.empty[gi]) { scores[idx] = -1.0f; return; }     // :145, :181-187
    const float* p = parts + (size_t)idx * 4;
    const int slot = q.tex_slot[qi];
    const float tex = p[3];
    const float a0 = slot == 0 ? tex : p[0];
    const float a1 = slot == 1 ? tex : p[1];
    const float a2 = slot == 2 ? tex : p[2];
    const float a28 = slot == 28 ? tex : 0.0f;
    float f = a0 + a1;
    f = f + a2;
    scores[idx] = (float)((double)f + (double)a28 * 0.3);
}

hipError_t read_phase_cycles(unsigned long long* out32, bool reset)
{
#ifdef AFIS_PHASE_TIMING
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_phase_cycles), 32 * sizeof(u64));
    if (e != hipSuccess) return e;
    if (reset) { u64 z[32] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)); }
    return e;
#else
    for (int i = 0; i < 32; ++i) out32[i] = 0;
    return hipSuccess;
#endif
}

hipError_t launch_fuse(const QueryDev& q, const GalleryDev& g, const float* parts, float* scores, hipStream_t stream)
{
    const long long n = (long long)q.nq * g.G;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fuse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q, g, parts, scores);
    return hipGetLastError();
}

}  // namespace afis
