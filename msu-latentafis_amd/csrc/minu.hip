// minu.hip — candidate generation of the minutiae-template scorer and the final score fusion:
//   S1  descriptor similarity  simi = max(0, A * B^T)              (matching/matcher.cpp:440-452)
//   S2  normalisation          norm = s / (rowsum + colsum - s + 1e-6) (matcher.cpp:455-470)
//   S3  the 120 correspondences with the largest norm, in rank order (matcher.cpp:473-488)
//   S10 fusion                 final = score[0] + score[1] + score[2] + score[28]*0.3   (matcher.cpp:376-417, :188)
// The correspondence lists go to HBM (120 x 8 B per task) and are consumed by k_graph_minutiae (graph.hip).
//
// k_minu_cands_rt (fast path, MFMA): one 256-thread workgroup per rolled template, looping over the launch's latent lists.
// k_minu_cands (any shape; the fast path's fallback): one workgroup per (query, selected latent template, gallery template) task.
// Canonical arithmetic (see oracle/afis_oracle.cpp): descriptor dot products are k-ascending fmaf chains, row/column sums are
// index-ascending, the normalisation is evaluated in double exactly as the reference's expression promotes it.
#include "afis_device.h"
#include "stdsort_order.h"
#include <cstdlib>

namespace afis {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
typedef unsigned long long u64;

#ifdef AFIS_PHASE_TIMING
// Development aid (make PHASE_TIMING=1): per-phase cycle shares as seen by thread 0 of every workgroup.  The stopwatch values are
// accumulated in LDS and flushed with ONE global atomic per slot at the end of the kernel — a global atomic inside the loop
// would stall the loads that follow it and distort the very phases it measures.
__device__ u64 g_phase_cycles[32];
#define PHASE_DECL() __shared__ u64 ph_acc[32]; if (threadIdx.x < 32) ph_acc[threadIdx.x] = 0; __syncthreads()
#define PHASE_FLUSH() do { __syncthreads(); if (threadIdx.x < 32 && ph_acc[threadIdx.x]) atomicAdd(&g_phase_cycles[threadIdx.x], ph_acc[threadIdx.x]); } while (0)
#define PHASE_INIT() u64 ph_t0 = __builtin_readcyclecounter()
#define PHASE(i) do { if (threadIdx.x == 0) { const u64 ph_t1 = __builtin_readcyclecounter(); ph_acc[i] += ph_t1 - ph_t0; ph_t0 = ph_t1; } } while (0)
#else
#define PHASE_DECL() do {} while (0)
#define PHASE_FLUSH() do {} while (0)
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
    uint16_t order[kFastN];         // option s3_tie_order 1: the element indices std::sort permutes (16 KB)
    uint16_t lpos[kFastN], rpos[kFastN];   // ... the two pointers' stops of a partition (stdsort_order.h: the closed form)
    uint16_t leaf_f[256], leaf_l[256];     // ... the ranges of <= 16 elements the introsort loop leaves for the final insertion sort
    int sort_stack[3 * 64];
};

// stdsort_prefix (stdsort_order.h) by ONE WAVE: the same ranges, the same pivots, the same swaps — a range's partition in its closed form (sso_partition_pivot_closed) with
// ballots and prefix counts for the loops, the final insertion sort one lane per leaf range (no element crosses a leaf's boundary: everything to its left is >= everything in it,
// and insertion moves an element left only past strictly smaller keys).  The depth-limit branch (heap sort) and the median of three stay with lane 0.
#define SSO_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
__device__ __forceinline__ int sso_lane_prefix(u64 mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0)); }
template <typename I> __device__ int sso_partition_wave(const SsoCtx<I>& c, int f, int l, I* lpos, I* rpos)
{
    I* A = c.A;
    const int lane = threadIdx.x & 63;
    if (lane == 0) {                                                                // __move_median_to_first(first, first + 1, mid, last - 1)
        const int a = f + 1, b = f + (l - f) / 2, cc = l - 1;
        if (sso_before(c, A[a], A[b])) {
            if (sso_before(c, A[b], A[cc])) sso_swap(A, f, b);
            else if (sso_before(c, A[a], A[cc])) sso_swap(A, f, cc);
            else sso_swap(A, f, a);
        } else if (sso_before(c, A[a], A[cc])) sso_swap(A, f, a);
        else if (sso_before(c, A[b], A[cc])) sso_swap(A, f, cc);
        else sso_swap(A, f, b);
    }
    SSO_WSYNC();
    const uint32_t pk = c.key[A[f]];
    int nl = 0, nr = 0;                                                             // uniform
    for (int p0 = f + 1; p0 < l; p0 += 64) {
        const int p = p0 + lane;
        const bool in = p < l;
        const uint32_t k = in ? c.key[A[p]] : 0u;
        const bool sl = in && k <= pk, sr = in && k >= pk;
        const u64 ml = __ballot(sl), mr = __ballot(sr);
        if (sl) lpos[nl + sso_lane_prefix(ml)] = (I)p;
        if (sr) rpos[nr + sso_lane_prefix(mr)] = (I)p;
        nl += (int)__popcll(ml); nr += (int)__popcll(mr);
    }
    SSO_WSYNC();
    int m = 0;
    const int lim = nl < nr ? nl : nr;
    for (int i0 = 0; i0 < lim; i0 += 64) {                                          // the swaps happen while L_i < R_i: a prefix of the i (L ascends, R descends)
        const int i = i0 + lane;
        const bool ok = i < lim && lpos[i] < rpos[nr - 1 - i];
        const int cnt = (int)__popcll(__ballot(ok));
        m += cnt;
        if (cnt < 64) break;
    }
    for (int i = lane; i < m; i += 64) {
        const int pl = (int)lpos[i], pr = (int)rpos[nr - 1 - i];
        const I x = A[pl], y = A[pr];
        A[pl] = y; A[pr] = x;
    }
    int cut = m < nl ? (int)lpos[m] : 0x7fffffff;
    if (m > 0) { const int r = (int)rpos[nr - m]; cut = r < cut ? r : cut; }
    SSO_WSYNC();
    return cut;
}
template <typename I> __device__ void stdsort_prefix_wave(I* A, int n, int K, const uint32_t* key, int* stack, I* lpos, I* rpos, uint16_t* leaf_f, uint16_t* leaf_l)
{
    const SsoCtx<I> c{A, key};
    const int lane = threadIdx.x & 63;
    if (n < 2) return;
    if (K > n) K = n;
    int sp = 1, n_leaf = 0;
    if (lane == 0) { stack[0] = 0; stack[1] = n; stack[2] = 2 * sso_floor_log2(n); }
    SSO_WSYNC();
    while (sp > 0) {
        --sp;
        int f = stack[3 * sp], l = stack[3 * sp + 1], d = stack[3 * sp + 2];        // uniform: every lane reads the same words
        SSO_WSYNC();                                                                // (read before lane 0 may push over them)
        if (f >= K) continue;
        bool by_heap = false;
        while (l - f > 16) {
            if (d == 0) { if (lane == 0) sso_heap_sort(c, f, l); SSO_WSYNC(); by_heap = true; break; }
            --d;
            const int cut = sso_partition_wave(c, f, l, lpos, rpos);
            if (lane == 0) { stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; }
            ++sp;
            SSO_WSYNC();
            l = cut;
        }
        if (l > f && !by_heap && n_leaf < 256) {      // (a heap-sorted range is in order: the insertion sort would move nothing; every other leaf ends below 120 + 16)
            if (lane == 0) { leaf_f[n_leaf] = (uint16_t)f; leaf_l[n_leaf] = (uint16_t)l; } ++n_leaf; }
    }
    SSO_WSYNC();
    for (int b = lane; b < n_leaf; b += 64) {                                       // the final insertion sort, one lane per leaf range
        const int lf = leaf_f[b], ll = leaf_l[b];
        for (int i = lf + 1; i < ll; ++i) {
            const I val = A[i];
            int j = i;
            while (j > lf && sso_before(c, val, A[j - 1])) { A[j] = A[j - 1]; --j; }
            A[j] = val;
        }
    }
    SSO_WSYNC();
}

// Global scratch of one workgroup (pairs too large for the LDS fast path): simi[n] | keys[n] | rowsum[2048] | colsum[2048] (| order[n] | lpos[n] | rpos[n] with option s3_tie_order)
__global__ __launch_bounds__(kThreads) void k_minu_cands(QueryDev q, GalleryDev g, float* __restrict__ scratch, size_t scratch_per_wg,
                                                         MinuCand* __restrict__ cands, int32_t* __restrict__ cand_n,
                                                         const int32_t* __restrict__ fb /* NULL: every task; else fb[0] tasks listed in fb[1 ...] */,
                                                         unsigned long long* __restrict__ diag /* NULL, or the launch group's diagnostics row */,
                                                         int ref_tie_order /* option s3_tie_order: equal norms in libstdc++'s std::sort order (stdsort_order.h) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MinuSmem& sm = *reinterpret_cast<MinuSmem*>(smem_raw);
    float* gscr = scratch + (size_t)blockIdx.x * scratch_per_wg;
    int parity = 0;
    PHASE_DECL();
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    const int tid = threadIdx.x;
    const long long n_work = fb ? (long long)fb[0] : n_tasks;
    if (diag != nullptr && blockIdx.x == 0 && tid == 0) diag[kDiagFallback] = (unsigned long long)n_work;
    for (long long work = blockIdx.x; work < n_work; work += gridDim.x) {
        const long long task = fb ? (long long)fb[1 + work] : work;
        // task order: gallery template fastest, then selected template, then query
        const int gi = (int)(task % g.G);
        const int qs = (int)(task / g.G);                    // qi*3 + s
        const int l0 = q.lm_off[qs], nL = q.lm_off[qs + 1] - l0;
        const int r0 = g.minu_off[gi], nR = g.minu_off[gi + 1] - r0;
        if (nL <= 0 || nR <= 0) { if (tid == 0) cand_n[task] = 0; continue; }     // matcher.cpp:400-404
        const int n = nL * nR;
        PHASE_INIT();
        const bool fast = nL <= kFastL && nR <= kFastR;
        const size_t half = (scratch_per_wg - 4096) / (ref_tie_order ? 5 : 2);     // afis_device.h::minu_scratch_floats(): option s3_tie_order adds three index arrays of n words
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
        if (ref_tie_order) {
            // The reference's own order of equal norms (matcher.cpp:473-476: std::sort of the indices 0 .. n-1 by norm, descending): every key of the task and the indices in
            // order, and ONE wave runs libstdc++'s algorithm on them as far as the first 120 positions need it (stdsort_order.h).  Up to 8192 similarities the arrays sit in LDS
            // (the keys in the GEMM's dead operand tiles; the matrix itself may be in global scratch), beyond that in the workgroup's global scratch with 32-bit indices.
            const bool in_lds = n <= kFastN;
            uint32_t* const keys32 = in_lds ? reinterpret_cast<uint32_t*>(sm.A) : gkeys;          // A and B are contiguous: 38.4 KB >= 4 n bytes (n <= 8192)
            static_assert(sizeof(sm.A) + sizeof(sm.B) >= sizeof(uint32_t) * kFastN && offsetof(MinuSmem, B) == sizeof(sm.A), "the keys reuse the GEMM's operand tiles");
            uint32_t* const gord = reinterpret_cast<uint32_t*>(gscr + 2 * half + 4096);
            if (in_lds) for (int e = tid; e < n; e += kThreads) { keys32[e] = norm_key(e); sm.order[e] = (uint16_t)e; }
            else        for (int e = tid; e < n; e += kThreads) { keys32[e] = norm_key(e); gord[e] = (uint32_t)e; }
            __syncthreads();
            if (tid < 64) {
                if (in_lds) stdsort_prefix_wave<uint16_t>(sm.order, n, topN, keys32, sm.sort_stack, sm.lpos, sm.rpos, sm.leaf_f, sm.leaf_l);
                else        stdsort_prefix_wave<uint32_t>(gord, n, topN, keys32, sm.sort_stack, gord + half, gord + 2 * half, sm.leaf_f, sm.leaf_l);
            }
            __syncthreads();
            if (tid < topN) {
                const int e = in_lds ? (int)sm.order[tid] : (int)gord[tid];
                const int i1 = e / nR, i2 = e - i1 * nR;
                MinuCand c; c.sim = simi[e]; c.li = (short)i1; c.ri = (short)i2;
                cands[(size_t)task * kTopMinu + tid] = c;
            }
            if (tid == 0) cand_n[task] = topN;
            __syncthreads();
            continue;
        }
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
    PHASE_FLUSH();
}


// ---------------------------------------------------------------------------------------------------------------------
// Fast path, rolled-template-stationary, in three SHAPE CLASSES (template parameter S = 1, 2, 4: workgroups of 256 S threads).
// One workgroup owns ONE rolled minutiae template at a time and runs every (latent, selected template) list of the launch that
// belongs to its class against it:
//   * S1, the one dense contraction of the path (the reference's Eigen GEMM), runs on the matrix cores with the exact-fp32
//     v_mfma_f32_16x16x4_f32.  A wave keeps the B fragment (16 rolled descriptors, k-permuted) of one column tile in 24
//     registers for all tasks of the rolled template; only the latent A fragments are fetched per task (shared by every
//     workgroup: L2/L1 resident).  Rolled descriptors therefore cross HBM -> CU once per launch instead of once per task.
//     With fewer column tiles than waves the waves of a column tile split the row tiles; column tiles beyond the wave count
//     are dealt round-robin over the waves with the B fragment fetched per item (an L1/L2 hit).
//   * the similarity matrix sits in LDS with an ODD row stride, so the sequential row sums (one lane per row) and the column
//     sums (one lane per column) are both bank-conflict free.
//   * S3: a 256-bin LDS histogram (16 bins per octave from 2^-15 up) of fp32 APPROXIMATIONS of
//     the norm keys — v_rcp_f32 instead of the reference's double division, at most 6 ulp away (see below) — finds the bin B
//     holding the 120th largest key.  Keys of zero similarities (half of all entries, all in one bin: they serialised the LDS
//     atomics) are not counted.  Only the ~130-190 entries that can still be among the 120 largest get the exact double-precision
//     key; they are ranked among themselves by counting on 48-bit composites (norm key, lowest element index first), which
//     yields exactly the reference's top 120 in the reference's order.  The kernel is bound by VALU issue, so the passes are laid
//     out for few instructions per element (thread = column x row phase; <= 32 keys per thread stay in registers between the passes).
//   * the classes: what bounds a task is its similarity matrix in LDS and the 32 keys per thread.  S = 1 (round 2's kernel): <= 64 latent x
//     <= 128 rolled minutiae, 37.7 KB, four workgroups per CU; S = 2: <= 16 384 similarities (128 x 128, 64 x 256 ...), 73 KB, two per CU;
//     S = 4: <= 512 rolled minutiae, <= 38 912 similarities incl. the padding column (151 x 256, 97 x 400, 75 x 512, 256 x 150 ...), 159 KB, one 16-wave workgroup per CU; its threads keep 32 keys each
//     and recompute the keys of the rows beyond (a second key block).  rt_max_rows(S, nR) (afis_device.h) is the
//     rule; a task goes to the smallest class that takes it (k_minu_classify lists, per class, the rolled templates that have such tasks in
//     this launch).
//   * anything else — more than 256 latent / 512 rolled minutiae or more than 38 912 similarities (the reference's reader allows 2000 per
//     template, matcher.cpp:788-790), fewer than 512 entries, a threshold in the two lowest bins (fewer than 120 similarities with a
//     norm of at least 2^-15), more than 256 candidates — is appended to a fallback list that k_minu_cands (exact threshold
//     search on exact keys, any shape) works off afterwards.
// Approximation bound.  a = sv * rcp(f + 1e-6f) with f the reference's own float (rowsum + colsum) - sv: the float sum
// f + 1e-6f (<= 2^-23 relative incl. the constant's rounding), v_rcp_f32 (1 ulp) and the product (2^-24) put a within
// 2.5 * 2^-23 relative of the exact quotient, whose own rounding to float adds 2^-24: at most 6 ulp between approximate and
// exact float key, |ka - ke| <= E = 8 as ordered integers.  With Ta = the 120th largest approximate key (in bin B): at least
// 120 exact keys are >= Ta - E, so every entry of the exact top 120 has ke >= Ta - E, hence ka >= Ta - 2E >= edge(B) - 2E.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSelBins = 256;                        // histogram bins: 16 per octave (4 mantissa bits) from 2^-15 up
constexpr int kBinBase = (127 - 15) * 16 - 1;        // key bits 30..19 of 2^-15, minus one: bin 0 = everything below (never counted)
constexpr int kHistMinN = 512;                       // smaller pairs go to the fallback kernel
constexpr int kCandCap = 256;                        // entries that may still be in the top 120 (exact keys computed for these)
constexpr uint32_t kKeySlack = 16;                   // 2E
#ifndef AFIS_PF_TILES
#define AFIS_PF_TILES 4
#endif
constexpr int kPfTiles = AFIS_PF_TILES;              // latent row tiles of the next task copied ahead into the tail of simi[] (6 KB each): at most this many, as many as the tail holds
template <int S> struct RtCfg {
    static constexpr int kT = 256 * S;               // threads of a workgroup
    static constexpr int kW = 4 * S;                 // its waves: resident column tiles
    static constexpr int kMaxR = rt_class_max_rolled(S);     // column sums: thread j; row sums: thread kMaxR + i
    static constexpr int kMaxL = rt_class_max_latent(S);
    static constexpr int kSimi = rt_class_simi_floats(S);
};
template <int S> struct __attribute__((aligned(16))) RtSmem {
    float simi[RtCfg<S>::kSimi];                     // row stride ld (odd, except the 128 of S = 1's widest templates)
    float rowsum[RtCfg<S>::kMaxL];
    float colsum[RtCfg<S>::kMaxR];
    uint32_t hist[kSelBins];                         // bin b >= 1: approximate keys with bits 30..19 == kBinBase + b (top bin: and above)
    u64 cand[kCandCap + 8];                          // exact composite keys of the candidates (+ zero padding: the ranking reads eight at a time)
    uint32_t cand_e[kCandCap];                       // their (bin << 17 | row << 9 | column)
    int pad_[4];
    int thr_bin, ticket;
    uint32_t sink[64];                               // where the histogram adds of entries that no bin counts go (one word per lane: conflict free): no branch around the atomic
};

__device__ __forceinline__ int lane_prefix(u64 mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0)); }

__device__ __forceinline__ uint32_t exact_norm_key(float sv, float rs, float cs)
{
    float f = rs + cs;
    f = f - sv;
    return ord_f32((float)((double)sv / ((double)f + 0.000001)));                               // matcher.cpp:467
}
__device__ __forceinline__ uint32_t approx_norm_key(float sv, float rs, float cs)
{
    float f = rs + cs;
    f = f - sv;
    const float a = sv * __builtin_amdgcn_rcpf(f + 0.000001f);
    return __float_as_uint(a) | 0x80000000u;                                                    // a >= 0: ord_f32's key
}

// Which rolled templates have work for which class in THIS launch (the class of a task follows from its two minutiae counts; the latent counts
// change from launch to launch).  One thread per rolled template: work[c * G + ...] receives the templates with at least one list in class c
// (ctl[c] = how many), in no particular order — every task writes its own slot of cands[] / cand_n[], so the order changes no result.  Tasks
// no class takes are settled here: no minutiae on a side -> an empty list (matcher.cpp:400-404); too large for every class -> the fallback list.
// ctl[0..2] = list lengths, ctl[4..6] = the classes' draw counters (zeroed by the launcher).
__global__ __launch_bounds__(256) void k_minu_classify(QueryDev q, GalleryDev g, int32_t* __restrict__ work, int32_t* __restrict__ ctl,
                                                       int32_t* __restrict__ cand_n, int32_t* __restrict__ fb)
{
    __shared__ int cnt[258];                                                         // after the scan: cnt[x], 1 <= x <= 256: lists with 1 <= nL <= x; cnt[257]: all non-empty lists; cnt[0]: the empty ones
    const int tid = threadIdx.x, lane = tid & 63;
    const int nqs = q.nq * 3;
    for (int i = tid; i < 258; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int qs = tid; qs < nqs; qs += 256) { const int nL = q.lm_off[qs + 1] - q.lm_off[qs]; atomicAdd(&cnt[nL <= 0 ? 0 : min(nL, 257)], 1); }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int x = 1; x <= 257; ++x) { run += cnt[x]; cnt[x] = run; } }
    __syncthreads();
    const int gi = blockIdx.x * 256 + tid;
    const bool live = gi < g.G;
    const int nR = live ? g.minu_off[gi + 1] - g.minu_off[gi] : 0;
    const int L1 = rt_max_rows(1, nR), L2 = rt_max_rows(2, nR), L4 = rt_max_rows(4, nR);
    // a class that takes nothing against this nR has L == 0, and cnt[0] is the count of EMPTY lists, not a prefix sum: its share is zero lists
    const int c1 = live && L1 > 0 ? cnt[L1] : 0, c2 = live && L2 > 0 ? cnt[L2] : c1, c4 = live && L4 > 0 ? cnt[L4] : c2;
    const int has[3] = {c1, c2 - c1, c4 - c2};                                       // nR <= 0: all three zero (the template is never listed)
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                                    // one atomic per wave and class
        const u64 m = __ballot(has[c] > 0);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&ctl[c], __popcll(m));
            base = __shfl(base, leader);
            if (has[c] > 0) work[(size_t)c * g.G + base + lane_prefix(m)] = gi;
        }
    }
    if (live && (nR <= 0 || cnt[0] > 0 || cnt[257] > c4)) {                     // some task of this rolled template belongs to no class
        for (int qs = 0; qs < nqs; ++qs) {
            const long long task = (long long)qs * g.G + gi;
            const int nL = q.lm_off[qs + 1] - q.lm_off[qs];
            if (nR <= 0 || nL <= 0) cand_n[task] = 0;
            else if (nL > L4) { const int p = atomicAdd(&fb[0], 1); fb[1 + p] = (int32_t)task; }
        }
    }
}

// fb[0] = number of fallback tasks, fb[1 ...] = their task indices
#ifndef AFIS_MC_ABLATE
#define AFIS_MC_ABLATE 0
#endif
#if AFIS_MC_ABLATE == 1                                                  // timing experiment only (wrong results): no workgroup barriers inside a task
#define RT_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define RT_SYNC() __syncthreads()
#endif
template <int S, int ref_tie_order /* option s3_tie_order: lists short of 120 positive norms, and lists in which positive norms tie, go to the any-shape kernel, which orders equal norms as std::sort does; an instantiation of its own: the default kernel carries none of it */>
__global__ __launch_bounds__(256 * S, 4) void k_minu_cands_rt(QueryDev q, GalleryDev g, const float4* __restrict__ lat_frag,
                                                                const float4* __restrict__ rol_frag,  // descriptors as operand fragments
                                                                MinuCand* __restrict__ cands, int32_t* __restrict__ cand_n, int32_t* __restrict__ fb,
                                                                const int32_t* __restrict__ work /* rolled templates with tasks of this class */,
                                                                int32_t* __restrict__ ctl /* ctl[c]: entries of work[]; ctl[4 + c]: the draw counter */,
                                                                unsigned long long* __restrict__ diag /* NULL, or the launch group's diagnostics row (afis_device.h) */)
{
    typedef RtCfg<S> Cfg;
    constexpr int kT = Cfg::kT, kW = Cfg::kW, kSimi = Cfg::kSimi, kCls = S == 1 ? 0 : S == 2 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char rt_smem_raw[];
    RtSmem<S>& sm = *reinterpret_cast<RtSmem<S>*>(rt_smem_raw);
    PHASE_DECL();
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nqs = q.nq * 3;
    auto to_fallback = [&](long long task) { const int p = atomicAdd(&fb[0], 1); fb[1 + p] = (int32_t)task; };
    // The clock the chip holds under this kernel (afis_timing.cands_clock_ghz): one lane of the first workgroups reads the shader-cycle counter and the constant
    // 100 MHz counter when it starts and when it leaves; the sums of the two differences go to the diagnostics row.
    const bool sampler = diag != nullptr && blockIdx.x < 8 && wave == 0;               // wave-uniform: the two counters stay in scalar registers
    unsigned long long clk0 = 0, wall0 = 0;
    if (sampler) { clk0 = __builtin_readcyclecounter(); wall0 = wall_clock64(); }
    int n_done = 0;                                                                 // tasks this workgroup completed (uniform)
    const int n_work = ctl[kCls];
    // A short work list (a class that only a few rolled templates reach in this launch: the medium class at the headline shapes gets ~50 of 100 000) would leave most of the chip idle
    // while a handful of workgroups walk 60 latent lists each: the lists of a template are then split over n_split tickets (uniform: from n_work, the grid and the list count).
    const int n_split = n_work <= 0 || n_work >= 2 * (int)gridDim.x ? 1 : min(nqs, (2 * (int)gridDim.x + n_work - 1) / n_work);
    const int n_tickets = n_work * n_split;
    // Rolled templates are DRAWN from a counter, not dealt by stride: the kernel may start on the part of the chip the (CU-masked) bound pass leaves free and spread
    // over the rest when that finishes (afis_search.cpp, option bound_cus): workgroups that start late must not find a fixed share of the work waiting for them.
    if (tid == 0) sm.pad_[1] = 0;                                                    // option s3_tie_order: "this list holds equal norms" (set in the ranking, consumed after the task's last barrier)
    for (;;) {
        if (tid == 0) sm.ticket = atomicAdd(&ctl[4 + kCls], 1);
        __syncthreads();
        const int wi = sm.ticket;
        __syncthreads();                                                            // everyone has read the ticket before thread 0 draws the next one
        if (wi >= n_tickets) break;
        const int gi = work[(size_t)kCls * g.G + wi / n_split];
        const int tpart = wi - (wi / n_split) * n_split;
        const int qs_lo = (int)((long long)nqs * tpart / n_split), qs_hi = (int)((long long)nqs * (tpart + 1) / n_split);   // this ticket's latent lists
        const int r0 = g.minu_off[gi], nR = g.minu_off[gi + 1] - r0;
        const int Lhi = rt_max_rows(S, nR), Llo = S == 1 ? 0 : rt_max_rows(S / 2, nR); // this class: Llo < nL <= Lhi (k_minu_classify listed the template: Lhi > 0)
        const int n_jt = (nR + 15) >> 4;
        const int ld = rt_row_stride(S, nR);
        const int R = kT / nR;                                                      // selection: row phases per column (>= 2)
        const int cr = tid / nR, cj = tid - cr * nR;                                // this thread's row phase and column (idle if cr >= R)
        // ---- the wave's resident B fragment: rolled descriptors of column tile jt_res (lane l: descriptor l&15, k-group l>>4); with fewer column tiles than
        // waves, P waves share a column tile and split its row tiles ----
        const int n_res = min(n_jt, kW), P = kW / n_res;
        const int jt_res = wave % n_res, part = wave / n_res;                       // part >= P: no resident work
        // fragment addresses = a wave-uniform base (scalar registers) + the lane's 16-byte slot as an unsigned 32-bit offset: no 64-bit per-lane pointers to keep alive
        const unsigned lane16 = (unsigned)lane * 16u;
        const char* const btiles = reinterpret_cast<const char*>(rol_frag + (size_t)g.minu_tile_off[gi] * (6 * 64));
        auto frag_at = [&](const char* base, int t, int v) -> const float4& { return *reinterpret_cast<const float4*>(base + ((unsigned)((t * 6 + v) * 1024) + lane16)); };
        float bres[24];
#pragma unroll
        for (int v = 0; v < 6; ++v) { const float4 x = frag_at(btiles, jt_res, v); bres[4 * v] = x.x; bres[4 * v + 1] = x.y; bres[4 * v + 2] = x.z; bres[4 * v + 3] = x.w; }
        auto next_in_class = [&](int from) {                                        // uniform: scalar loads
            int x = from;
            for (; x < qs_hi; ++x) { const int nl = q.lm_off[x + 1] - q.lm_off[x]; if (nl > Llo && nl <= Lhi) break; }
            return x;
        };
        // The latent row tiles of the NEXT task of this rolled template are copied into the unused tail of simi[] while this task is being selected from
        // (global_load_lds: memory -> LDS without registers; one 16-byte element per lane, one copy for the workgroup's waves, which each fetched every tile
        // themselves before) — the fetch was the exposed L2 round trip at the head of every task.  pf_qs: the task whose tiles the tail holds (-1: none).
        int pf_qs = -1, pf_n = 0;
        int qs_next = next_in_class(qs_lo);
        for (int qs = qs_next; qs < qs_hi; qs = qs_next) {
            qs_next = next_in_class(qs + 1);
            const long long task = (long long)qs * g.G + gi;
            const int l0 = q.lm_off[qs], nL = q.lm_off[qs + 1] - l0;
            const int n = nL * nR;
            if (n < kHistMinN) { if (tid == 0) to_fallback(task); continue; }
            PHASE_INIT();
            // ---- S1 (matcher.cpp:440-452): simi = max(0, A * B^T); v_mfma_f32_16x16x4_f32 == the k-ascending fmaf chain ----
            // Operand layout: lane l supplies A[i = l&15][k = 4s + (l>>4)] and B[k][j = l&15] at step s = 4v + c; the fragment arrays
            // hold exactly that per (tile, v, lane), so a fragment is six fully coalesced 1 KB loads.
            const int n_it = (nL + 15) >> 4;
            const char* const atiles = reinterpret_cast<const char*>(lat_frag + (size_t)q.lm_tile_off[qs] * (6 * 64));
            const int n_pf = pf_qs == qs ? pf_n : 0;                                // uniform: row tiles 0 .. n_pf - 1 wait in the tail of simi[], tile t in the t-th 6 KB from the end
            const char* const a_end = reinterpret_cast<const char*>(sm.simi + kSimi);
            auto a_lds = [&](int t) { return a_end - (t + 1) * (6 * 64 * 16); };
            auto load_frag = [&](const char* __restrict__ tiles, int t, float (&f)[24]) {
#pragma unroll
                for (int v = 0; v < 6; ++v) { const float4 x = frag_at(tiles, t, v); f[4 * v] = x.x; f[4 * v + 1] = x.y; f[4 * v + 2] = x.z; f[4 * v + 3] = x.w; }
            };
            auto store_tile = [&](int it, int jt, const f32x4& acc) {              // D: col = lane & 15, row = (lane >> 4) * 4 + r
                const int j = jt * 16 + li;
                const int i0 = it * 16 + lg * 4;
                float* const p = &sm.simi[(int)__umul24((unsigned)i0, (unsigned)ld) + j];   // one 24-bit multiply per tile; the four rows are ld floats apart
                if (j < nR) {                                                       // (a column beyond the template would land in the next row)
                    if (it * 16 + 16 <= nL) {                                       // uniform: a full row tile needs no row tests
#pragma unroll
                        for (int r = 0; r < 4; ++r) { float v = acc[r]; if (v < 0) v = 0; p[r * ld] = v; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { float v = acc[r]; if (v < 0) v = 0; if (i0 + r < nL) p[r * ld] = v; }
                    }
                }
            };
            // The six loads of a fragment are issued together (sched_barrier pins that: left alone, the scheduler sinks every load next
            // to its first use — one exposed memory latency per 4 MFMAs instead of one per tile).
            auto mfma_tile = [&](const float (&af)[24], const float (&bf)[24]) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 24; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st], bf[st], acc, 0, 0, 0);
                return acc;
            };
            if (part < P) {                                                         // column tile jt_res, this wave's share of the row tiles: B stays in registers
                // row tiles in pairs: two independent accumulator chains keep the matrix pipe issuing every 32 cycles (one chain alone
                // waits 40+ cycles for each result)
                for (int it = 2 * part; it < n_it; it += 2 * P) {
                    float a0[24], a1[24];
                    if (it < n_pf) load_frag(a_lds(it), 0, a0); else load_frag(atiles, it, a0);
                    if (it + 1 < n_it) {
                        if (it + 1 < n_pf) load_frag(a_lds(it + 1), 0, a1); else load_frag(atiles, it + 1, a1);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int st = 0; st < 24; ++st) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[st], bres[st], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[st], bres[st], acc1, 0, 0, 0);
                        }
                        store_tile(it, jt_res, acc0);
                        store_tile(it + 1, jt_res, acc1);
                    } else {
                        __builtin_amdgcn_sched_barrier(0);
                        store_tile(it, jt_res, mfma_tile(a0, bres));
                    }
                }
            }
            for (int item = wave; item < (n_jt - kW) * n_it; item += kW) {           // column tiles kW ..: dealt round-robin
                const int jt = kW + item / n_it, it = item - (jt - kW) * n_it;
                float af[24], bf[24];
                load_frag(btiles, jt, bf);
                if (it < n_pf) load_frag(a_lds(it), 0, af); else load_frag(atiles, it, af);
                __builtin_amdgcn_sched_barrier(0);
                store_tile(it, jt, mfma_tile(af, bf));
            }
            if (S == 1 || tid < kSelBins) sm.hist[tid] = 0u;
            if (tid == 0) sm.thr_bin = -1;
            RT_SYNC();
            pf_qs = -1;
            if (qs_next < qs_hi) {                                                   // uniform
                const int nLn = q.lm_off[qs_next + 1] - q.lm_off[qs_next];
                // the tail must clear this task's matrix (still being read) and the next one's (written before the tiles are read): as many tiles as fit above both
                const int n_cp = min(min((nLn + 15) >> 4, kPfTiles), (kSimi - max(nL, nLn) * ld) / (6 * 64 * 4));
                if (n_cp > 0) {
                    const float4* src = lat_frag + (size_t)q.lm_tile_off[qs_next] * (6 * 64) + lane;
                    float4* const end4 = reinterpret_cast<float4*>(sm.simi + kSimi);
                    for (int c = wave; c < n_cp * 6; c += kW) {                      // chunk = 64 lanes x 16 B = one (tile, v) slice
                        const int t = c / 6, v = c - t * 6;
                        __builtin_amdgcn_global_load_lds(src + c * 64, (__attribute__((address_space(3))) void*)(end4 - (t + 1) * (6 * 64) + v * 64), 16, 0, 0);
                    }
                    pf_qs = qs_next; pf_n = n_cp;
                }
            }
            PHASE(16);
#if AFIS_MC_ABLATE == 2                                                  // timing experiment only: the GEMM and its stores, no selection
            if (tid == 0) cand_n[task] = 0;
            __syncthreads();
            continue;
#endif
            // ---- S2 (:455-456): index-ascending sums; odd row stride: both walks are conflict free.  The adds are one dependent chain per lane (that IS the reference's order); the
            // LDS reads are not: eight are issued before the eight adds that use them, and the NEXT eight are already in flight while those adds run.
            auto seq_sum = [&](const float* __restrict__ p, const int stride, const int n) {
                float sacc = 0.f;
                int k = 0;
                if (n >= 8) {
                    float cur[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cur[u] = p[u * stride];
                    for (; k + 16 <= n; k += 8) {
                        float nxt[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) nxt[u] = p[(k + 8 + u) * stride];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 8; ++u) sacc += cur[u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sacc += cur[u];
                    k += 8;
                }
                for (; k < n; ++k) sacc += p[k * stride];
                return sacc;
            };
            if (tid < nR) sm.colsum[tid] = seq_sum(&sm.simi[tid], ld, nL);
            else if (tid >= Cfg::kMaxR && tid - Cfg::kMaxR < nL) sm.rowsum[tid - Cfg::kMaxR] = seq_sum(&sm.simi[(tid - Cfg::kMaxR) * ld], 1, nR);
            RT_SYNC();
            PHASE(17);
            // ---- S3 (:461-488): the 120 largest norm values.  The kernel is bound by VALU issue (about 2000 wave-instructions per wave
            // and task before this layout), so the selection is organised for few instructions per element: thread = (column cj, row
            // phase cr) walks the rows cr, cr + R, ... of ITS column — the column sum stays in a register, the address advances by a
            // constant — and keeps the approximate keys in registers for the second pass.
            const int n_rows = (nL + R - 1) / R;                                     // <= 32 (S = 4: <= 64, the rows beyond 32 R form a second key block): rt_max_rows()
            const int my_rows = cr < R ? (nL - cr + R - 1) / R : 0;
            uint32_t rk[32];
            {
                const float cs = sm.colsum[cj];
                int a = cr * ld + cj, i = cr;
                const int a_step = R * ld;
                // four rows per trip: their eight LDS reads are issued together and nothing branches around an element (a row beyond this thread's share reads element 0 and
                // gets key 0, which no bin counts) — one element at a time, each with its own wait and its own exec-mask regions, made this pass 20 % of the kernel
#pragma unroll
                for (int tb = 0; tb < 32; tb += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) rk[tb + u] = 0u;
                    if (tb < n_rows) {                                               // uniform
                        float sv[4], rs[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool ok = tb + u < my_rows;
                            sv[u] = sm.simi[ok ? a + u * a_step : 0]; rs[u] = sm.rowsum[ok ? i + u * R : 0];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t key = approx_norm_key(sv[u], rs[u], cs) & (uint32_t)-(int)(tb + u < my_rows);   // (a mask, not a select: the compiler turns the select into a branch region)
                            rk[tb + u] = key;
                            const int bin = min((int)((key >> 19) & 0xfffu) - kBinBase, kSelBins - 1);
                            atomicAdd(bin > 0 ? &sm.hist[bin] : &sm.sink[lane], 1u);  // zero similarities, norms < 2^-15 and absent rows are not counted
                        }
                        a += 4 * a_step; i += 4 * R;
                    }
                }
                if (S == 4 && n_rows > 32) {                                         // second key block (large class only): counted now, recomputed in the candidate pass instead of kept
                    for (int t = 32; t < n_rows; ++t) {                              // uniform
                        if (t < my_rows) {
                            const uint32_t key = approx_norm_key(sm.simi[a], sm.rowsum[i], cs);
                            const int bin = min((int)((key >> 19) & 0xfffu) - kBinBase, kSelBins - 1);
                            if (bin > 0) atomicAdd(&sm.hist[bin], 1u);
                        }
                        a += R * ld; i += R;
                    }
                }
            }
            RT_SYNC();
            PHASE(29);
            if (wave == 0) {   // one wave scans the 256 bins (lane = four consecutive bins): suffix sums over the higher bins find the bin holding the 120th largest approximate key
                const uint4 h = reinterpret_cast<const uint4*>(sm.hist)[lane];
                const int s0 = (int)(h.x + h.y + h.z + h.w);
                int suf = s0;                                                        // becomes the sum over this lane and every higher one: four row_shl steps inside the rows of 16 lanes ...
                suf += __builtin_amdgcn_update_dpp(0, suf, 0x101, 0xf, 0xf, true);
                suf += __builtin_amdgcn_update_dpp(0, suf, 0x102, 0xf, 0xf, true);
                suf += __builtin_amdgcn_update_dpp(0, suf, 0x104, 0xf, 0xf, true);
                suf += __builtin_amdgcn_update_dpp(0, suf, 0x108, 0xf, 0xf, true);
                const int r1 = __builtin_amdgcn_readlane(suf, 16), r2 = __builtin_amdgcn_readlane(suf, 32), r3 = __builtin_amdgcn_readlane(suf, 48);   // ... and the totals of the rows above
                suf += lane < 16 ? r1 + r2 + r3 : lane < 32 ? r2 + r3 : lane < 48 ? r3 : 0;
                const int a3 = suf - s0, a2 = a3 + (int)h.w, a1 = a2 + (int)h.z, a0 = a1 + (int)h.y;     // entries in the bins ABOVE each of the four
                if (a3 < kTopMinu && a3 + (int)h.w >= kTopMinu) sm.thr_bin = 4 * lane + 3;
                if (a2 < kTopMinu && a2 + (int)h.z >= kTopMinu) sm.thr_bin = 4 * lane + 2;
                if (a1 < kTopMinu && a1 + (int)h.y >= kTopMinu) sm.thr_bin = 4 * lane + 1;
                if (a0 < kTopMinu && a0 + (int)h.x >= kTopMinu) sm.thr_bin = 4 * lane;
                reinterpret_cast<uint4*>(sm.hist)[lane] = make_uint4((uint32_t)a0, (uint32_t)a1, (uint32_t)a2, (uint32_t)a3);   // from here on: where the next candidate of each bin goes
            }
            RT_SYNC();
            PHASE(30);
            const int Braw = __builtin_amdgcn_readfirstlane(sm.thr_bin);            // (uniform: one LDS word)
            // Fewer than 120 positive norms, every one of them at least 2^-15 (a latent and a rolled print whose descriptors point away from each other: nearly every similarity is
            // clamped to zero, matcher.cpp:447-451 — 8 % of the pairs of bench.py --workload structured, which the any-shape kernel did at 30 x the time): the positive entries are all
            // candidates and are ranked as always; the rest of the 120 are zero entries, which tie, in ascending element order (tie rule) — filled in below.
            bool fill = false;
            if (__builtin_expect(Braw < 0 && !ref_tie_order, 0)) {                                        // uniform, rare: are there positive norms below 2^-15 (uncounted, bin <= 0)?  Looked for only here — the histogram pass pays nothing for it
                if (tid == 0) sm.pad_[0] = 0;
                RT_SYNC();
                bool tiny = false;
                {   // (the keys are recomputed from the matrix, not read from the 32 registers that hold them: a rare path that walks the register array costs the common one 1 % — measured)
                    const float cs2 = my_rows > 0 ? sm.colsum[cj] : 0.0f;
                    for (int t = 0; t < my_rows; ++t) {
                        const int i = cr + R * t;
                        const uint32_t key = approx_norm_key(sm.simi[i * ld + cj], sm.rowsum[i], cs2);
                        tiny |= (key & 0x7fffffffu) != 0u && (int)((key >> 19) & 0xfffu) - kBinBase <= 0;
                    }
                }
                if (tiny) sm.pad_[0] = 1;
                RT_SYNC();
                fill = __builtin_amdgcn_readfirstlane(sm.pad_[0]) == 0;
            }
            if (__builtin_expect(Braw < 2 && !fill, 0)) { if (tid == 0) to_fallback(task); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RT_SYNC(); continue; }   // a threshold in or next to bin 0, or tiny positive norms among fewer than 120
            const int B = fill ? 0 : Braw;
            // ---- a crowded threshold bin: a second histogram inside it ----
            // Descriptors of extracted prints lie near a common manifold: a pair's norm keys then crowd into an octave or less, and the threshold bin alone (1/16 octave) can hold
            // more entries than the candidate list (bench.py --workload structured: 8 % of the tasks went to the any-shape kernel for this reason, at 30 x the time).  The bin's keys
            // are counted again by their next 8 bits (bits 18..11: 256 sub-bins of 2048 ordered-key units, still >> 2E); the sub-bin B2 holding the 120th largest key moves the
            // candidate edge up to edge(B) + B2 * 2^11.  The argument of the header holds with Ta >= the new edge; the candidates' grouping by bin is unchanged (group B = the last).
            uint32_t edge = fill ? 0x80000001u + kKeySlack : 0x80000000u | ((uint32_t)(B + kBinBase) << 19);     // fill: every positive key
            if (!fill) {
                const int above = (int)sm.hist[B], inbin = (int)sm.hist[B - 1] - above;     // uniform (after the scan hist[b] = the entries in the bins above b)
                if (__builtin_expect(above + inbin > kCandCap - 32 && B < kSelBins - 1, 0)) {                 // (the top bin also holds everything above it: its keys' lower bits say nothing)
                    uint32_t* const h2 = reinterpret_cast<uint32_t*>(sm.cand);             // the composites' array is free until the candidates are keyed
                    if (tid < 256) h2[tid] = 0u;
                    RT_SYNC();
                    const uint32_t bin_bits = edge >> 19;
                    {   // (keys recomputed from the matrix: see the fill path above)
                        const float cs2 = my_rows > 0 ? sm.colsum[cj] : 0.0f;
                        for (int t = 0; t < my_rows; ++t) {
                            const int i = cr + R * t;
                            const uint32_t key = approx_norm_key(sm.simi[i * ld + cj], sm.rowsum[i], cs2);
                            if ((key >> 19) == bin_bits) atomicAdd(&h2[(key >> 11) & 255u], 1u);
                        }
                    }
                    RT_SYNC();
                    if (wave == 0) {                                                     // the scan of the first histogram, over the sub-bins: the (120 - above)-th largest key of the bin
                        const int need = kTopMinu - above;                               // 1 <= need <= inbin: exactly one sub-bin qualifies
                        if (lane == 0) sm.thr_bin = 0;                                   // (were none to qualify, the edge stays the bin's own)
                        const uint4 h = reinterpret_cast<const uint4*>(h2)[lane];
                        const int s0 = (int)(h.x + h.y + h.z + h.w);
                        int suf = s0;
                        suf += __builtin_amdgcn_update_dpp(0, suf, 0x101, 0xf, 0xf, true);
                        suf += __builtin_amdgcn_update_dpp(0, suf, 0x102, 0xf, 0xf, true);
                        suf += __builtin_amdgcn_update_dpp(0, suf, 0x104, 0xf, 0xf, true);
                        suf += __builtin_amdgcn_update_dpp(0, suf, 0x108, 0xf, 0xf, true);
                        const int r1 = __builtin_amdgcn_readlane(suf, 16), r2 = __builtin_amdgcn_readlane(suf, 32), r3 = __builtin_amdgcn_readlane(suf, 48);
                        suf += lane < 16 ? r1 + r2 + r3 : lane < 32 ? r2 + r3 : lane < 48 ? r3 : 0;
                        const int a3 = suf - s0, a2 = a3 + (int)h.w, a1 = a2 + (int)h.z, a0 = a1 + (int)h.y;
                        if (a3 < need && a3 + (int)h.w >= need) sm.thr_bin = 4 * lane + 3;
                        if (a2 < need && a2 + (int)h.z >= need) sm.thr_bin = 4 * lane + 2;
                        if (a1 < need && a1 + (int)h.y >= need) sm.thr_bin = 4 * lane + 1;
                        if (a0 < need && a0 + (int)h.x >= need) sm.thr_bin = 4 * lane;
                    }
                    RT_SYNC();
                    edge |= (uint32_t)__builtin_amdgcn_readfirstlane(sm.thr_bin) << 11;
                }
            }
            // ---- the candidates: approximate key >= edge - 2E ----
            {
                const uint32_t edge_s = edge - kKeySlack;                            // key + slack >= edge  <=>  key >= edge - slack (no wrap: keys of real entries are below 0xff800000, unused slots hold 0)
                uint32_t hits = 0;
#pragma unroll
                for (int t = 31; t >= 0; --t)                                        // hits = 2 hits + (key >= edge_s): a compare into the carry and an add-with-carry per key (the compiler's compare / select / shift-or took four)
                    asm("v_cmp_le_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(hits) : "v"(rk[t]), "s"(edge_s) : "vcc");
                // The candidates are appended GROUPED BY BIN, highest bin first (every key of a bin above B is a candidate, so the suffix
                // counts of the scan are the groups' start positions; keys of bin B - 1 within the slack join group B, the last one).
                // The bin of a hit is recomputed from LDS: indexing rk[] with a runtime t would put the 32 keys into scratch.
                const float cs = sm.colsum[cj];
                while (hits) {
                    const int t = __ffs(hits) - 1;
                    hits &= hits - 1;
                    const int i = cr + R * t;
                    const uint32_t key = approx_norm_key(sm.simi[i * ld + cj], sm.rowsum[i], cs);
                    const int bin = max(min((int)((key >> 19) & 0xfffu) - kBinBase, kSelBins - 1), B);
                    const uint32_t p = atomicAdd(&sm.hist[bin], 1u);
                    if (p < (uint32_t)kCandCap) sm.cand_e[p] = (uint32_t)((bin << 17) | (i << 9) | cj);
                }
                if (S == 4 && n_rows > 32) {                                         // the second key block's entries: keys recomputed
                    for (int t = 32; t < my_rows; ++t) {
                        const int i = cr + R * t;
                        const uint32_t key = approx_norm_key(sm.simi[i * ld + cj], sm.rowsum[i], cs);
                        if (key + kKeySlack >= edge) {
                            const int bin = max(min((int)((key >> 19) & 0xfffu) - kBinBase, kSelBins - 1), B);
                            const uint32_t p = atomicAdd(&sm.hist[bin], 1u);
                            if (p < (uint32_t)kCandCap) sm.cand_e[p] = (uint32_t)((bin << 17) | (i << 9) | cj);
                        }
                    }
                }
            }
            RT_SYNC();
            PHASE(31);
            const int n_c = (int)sm.hist[B];                                         // >= 120: group B ends the list
            if (n_c > kCandCap) { if (tid == 0) to_fallback(task); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RT_SYNC(); continue; }
            int ci = 0, cj2 = 0, cbin = 0;
            uint32_t ke = 0u;
            if (tid < n_c) {                                                         // one exact (double-precision) key per candidate
                const uint32_t pe = sm.cand_e[tid];
                cbin = (int)(pe >> 17); ci = (int)((pe >> 9) & 255u); cj2 = (int)(pe & 511u);
                ke = exact_norm_key(sm.simi[ci * ld + cj2], sm.rowsum[ci], sm.colsum[cj2]);
                sm.cand[tid] = ((u64)ke << 16) | (u64)(65535 - (ci * nR + cj2));
            }
            if (tid < 8) sm.cand[n_c + tid] = 0ull;                                  // padding: the ranking below reads the list eight composites at a time
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's share of the next task's tiles has landed in LDS (waited for here, before the list's stores join the counter)
            RT_SYNC();
            PHASE(18);
            // ---- rank the candidates by counting (composites are unique); ranks < 120 are the list, in the reference's order.
            // The list is grouped by the bin of the APPROXIMATE key, highest bin first; sm.hist[b] now holds the END of group b (= the start of group b - 1).  Approximate and
            // exact key differ by at most E = 8 << a bin's width (2^19): a candidate of bin g whose exact key lies at least 2E inside the bin is beaten by every candidate of the
            // higher bins and beats every one of the lower bins, so it counts the larger composites of its OWN group only (a third of the three-bin window every candidate
            // used to walk: the loop is a chain of LDS round trips as long as the wave's longest range); one within 2E of an edge also walks the group on that side.
            // The range is widened to even positions: the extra element in front is larger, the one behind smaller or the pad.
            if (tid < n_c) {
                const u64 mine = sm.cand[tid];
                const uint32_t edge_own = 0x80000000u | ((uint32_t)(cbin + kBinBase) << 19), edge_up = edge_own + (1u << 19);
                const bool need_up = cbin + 1 < kSelBins && ke + kKeySlack >= edge_up;   // (the top bin also holds everything above it: nothing lies higher)
                const bool need_dn = ke < edge_own + kKeySlack;
                const int lo = need_up ? (cbin + 2 < kSelBins ? (int)sm.hist[cbin + 2] & ~1 : 0) : (cbin + 1 < kSelBins ? (int)sm.hist[cbin + 1] & ~1 : 0);
                const int hi = cbin <= B ? n_c : need_dn ? (cbin - 1 > B ? (int)sm.hist[cbin - 1] : n_c) : (int)sm.hist[cbin];
                int r = lo;
                const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>(sm.cand);
                // eight composites per trip, all four reads in flight together (the loop is a chain of LDS round trips, as long as the longest range of the wave: two per trip made it
                // 22 % of the kernel).  What a trip reads beyond `hi` belongs to lower bins or is padding: smaller than `mine`, never counted.
                for (int k = lo; k < hi; k += 8) {
                    const ulonglong2 k0 = c2[(k >> 1)], k1 = c2[(k >> 1) + 1], k2 = c2[(k >> 1) + 2], k3 = c2[(k >> 1) + 3];
                    r += (int)(k0.x > mine) + (int)(k0.y > mine) + (int)(k1.x > mine) + (int)(k1.y > mine) + (int)(k2.x > mine) + (int)(k2.y > mine) + (int)(k3.x > mine) + (int)(k3.y > mine);
                }
                if (r < kTopMinu) {
                    MinuCand cd; cd.sim = sm.simi[ci * ld + cj2]; cd.li = (short)ci; cd.ri = (short)cj2;
                    cands[(size_t)task * kTopMinu + r] = cd;
                }
                if (ref_tie_order && r <= kTopMinu) sm.cand_e[r] = ke;               // (the option's instantiation only) the norm keys in rank order, one beyond the list: equal neighbours = equal POSITIVE norms in
                                                                                     // or at the end of the list, which libstdc++'s sort orders its own way (cand_e[] is dead: its entries were read before the composites were written)
            }
            if (fill && wave == 0) {                                                 // ranks n_c .. 119: the first zero similarities in element order (n >= 512 entries, fewer than 120 of them positive: there are enough)
                int rank = n_c;
                for (int e0 = 0; e0 < n && rank < kTopMinu; e0 += 64) {
                    const int e = e0 + lane;
                    const int i = e / nR, j = e - i * nR;
                    const bool z = e < n && sm.simi[i * ld + j] == 0.0f;
                    const u64 zm = __ballot(z);
                    const int r = rank + lane_prefix(zm);
                    if (z && r < kTopMinu) { MinuCand cd; cd.sim = 0.0f; cd.li = (short)i; cd.ri = (short)j; cands[(size_t)task * kTopMinu + r] = cd; }
                    rank += (int)__popcll(zm);
                }
            }
            if (tid == 0) cand_n[task] = kTopMinu;
            ++n_done;
#if AFIS_MC_ABLATE != 3                                                  // (3: timing experiment only — no barrier at the end of a task: what dropping it could give at most)
            RT_SYNC();
            if (ref_tie_order) {                                                     // one list in 10^5: two equal positive norms among the first 121 -> the any-shape kernel redoes the list in the reference's
                if (tid + 1 < min(n_c, kTopMinu + 1) && sm.cand_e[tid] == sm.cand_e[tid + 1]) sm.pad_[1] = 1;   // sort order (it runs after this kernel, on the same stream)
                RT_SYNC();
                if (tid == 0 && sm.pad_[1] != 0) { sm.pad_[1] = 0; to_fallback(task); }
            }
#endif
            PHASE(20);
        }
    }
    if (diag != nullptr && tid == 0) {
        if (n_done) atomicAdd(&diag[kDiagSmall + kCls], (unsigned long long)n_done);
        if (sampler) { atomicAdd(&diag[kDiagCandsClk], (unsigned long long)__builtin_readcyclecounter() - clk0); atomicAdd(&diag[kDiagCandsWall], (unsigned long long)wall_clock64() - wall0); }   // (tid 0 is in wave 0)
    }
    PHASE_FLUSH();
}

template <int S, int REF>
static hipError_t launch_rt_class_t(const QueryDev& q, const GalleryDev& g, MinuCand* cands, int32_t* cand_n, int32_t* fallback, int32_t* work, int32_t* ctl, unsigned long long* diag, hipStream_t stream)
{
    // opt-in to > 64 KB of dynamic LDS: a per-device function attribute, set on every launch (cheap) rather than cached per process
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_minu_cands_rt<S, REF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RtSmem<S>));
    if (e != hipSuccess) return e;
    const int per_cu = 4 / S;                                                      // 4 / 2 / 1 workgroups (16 waves) per CU, persistent: rolled templates are drawn from ctl[4 + class]
#ifndef AFIS_RT_GRID_CAP
#define AFIS_RT_GRID_CAP 0                                                         // occupancy probe (a build-time constant since round 6: 512 = two small-class workgroups per CU, ...)
#endif
    const int full = (AFIS_RT_GRID_CAP > 0 && S == 1) ? AFIS_RT_GRID_CAP : 256 * per_cu;
    const int grid = g.G < full ? g.G : full;
    hipLaunchKernelGGL((k_minu_cands_rt<S, REF>), dim3(grid), dim3(256 * S), sizeof(RtSmem<S>), stream, q, g, q.lm_frag, g.minu_frag, cands, cand_n, fallback, work, ctl, diag);
    return hipGetLastError();
}
template <int S>
static hipError_t launch_rt_class(const QueryDev& q, const GalleryDev& g, MinuCand* cands, int32_t* cand_n, int32_t* fallback, int32_t* work, int32_t* ctl, unsigned long long* diag, int ref_tie_order, hipStream_t stream)
{
    return ref_tie_order ? launch_rt_class_t<S, 1>(q, g, cands, cand_n, fallback, work, ctl, diag, stream) : launch_rt_class_t<S, 0>(q, g, cands, cand_n, fallback, work, ctl, diag, stream);
}

hipError_t launch_minu_cands(const QueryDev& q, const GalleryDev& g, float* scratch, size_t scratch_floats_per_wg, int n_wg,
                             int force_generic, MinuCand* cands, int32_t* cand_n, int32_t* fallback, int max_nL, int max_nR, unsigned long long* diag, hipStream_t stream)
{
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7ffffff0LL) return hipErrorInvalidValue;
    hipError_t e;
    const int ref_tie_order = (force_generic >> 1) & 1;                             // bit 1 of the flags word: option s3_tie_order
    force_generic &= 1;
    if (!force_generic) {
        int32_t* ctl = fallback + 1 + n_tasks;                                    // minu_fb_ints(): [count | n_tasks task ids | 8 control words | 3 G work-list entries]
        int32_t* work = ctl + 8;
        e = hipMemsetAsync(fallback, 0, sizeof(int32_t), stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(ctl, 0, 8 * sizeof(int32_t), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_minu_classify, dim3((g.G + 255) / 256), dim3(256), 0, stream, q, g, work, ctl, cand_n, fallback);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_rt_class<1>(q, g, cands, cand_n, fallback, work, ctl, diag, ref_tie_order, stream);
        if (e != hipSuccess) return e;
        // the larger classes only when the shapes of this launch can reach them (a kernel that finds its work list empty still costs a launch)
        const int mr = max_nR < 256 ? (max_nR > 0 ? max_nR : 1) : 256;
        if (max_nL > rt_class_max_latent(1) || max_nR > rt_class_max_rolled(1)) {
            e = launch_rt_class<2>(q, g, cands, cand_n, fallback, work, ctl, diag, ref_tie_order, stream);
            if (e != hipSuccess) return e;
            if (max_nR > rt_class_max_rolled(2) || max_nL > rt_max_rows(2, mr)) {
                e = launch_rt_class<4>(q, g, cands, cand_n, fallback, work, ctl, diag, ref_tie_order, stream);
                if (e != hipSuccess) return e;
            }
        }
    }
    // opt-in to > 64 KB of dynamic LDS: a per-device function attribute, set on every launch (cheap) rather than cached per process
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_minu_cands), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MinuSmem));
    if (e != hipSuccess) return e;
    const int grid = (int)(n_tasks < n_wg ? n_tasks : n_wg);
    // (the arrays of the std::sort restatement — 50 KB at the end of MinuSmem — are asked for only when the option is on: two workgroups per CU otherwise, as before)
    const size_t smem = ref_tie_order ? sizeof(MinuSmem) : offsetof(MinuSmem, order);
    hipLaunchKernelGGL(k_minu_cands, dim3(grid), dim3(kThreads), smem, stream, q, g, scratch, scratch_floats_per_wg, cands, cand_n,
                       force_generic ? (const int32_t*)nullptr : fallback, diag, ref_tie_order);
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

// =====================================================================================================================
// S11 rank list (matcher.cpp:306-309: indices sorted by score, descending; equal scores by ascending index — the documented
// tie rule).  One 1024-thread workgroup per query; k rounds of a workgroup-wide maximum over UNIQUE 64-bit composites
// (ordered score bits << 32 | ~index): round r finds the largest composite below round r-1's, so nothing is marked or moved.
// The shard's scores stay in HBM/L2 (k passes over G floats per query: 24 x 0.4 MB at G = 100k); only k x 12 bytes per query go
// back to the host.
// =====================================================================================================================
constexpr int kTopkThreads = 1024;
__global__ __launch_bounds__(kTopkThreads) void k_topk(const float* __restrict__ scores, int G, int k, long long index_base,
                                                      long long* __restrict__ out_idx, float* __restrict__ out_score)
{
    __shared__ u64 s_part[kTopkThreads / 64];
    __shared__ u64 s_best;
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = scores + (size_t)qi * G;
    u64 prev = ~0ull;
    for (int r = 0; r < k; ++r) {
        u64 best = 0;                                                     // every real composite is > 0 (ord_f32 >= 0x007fffff)
        for (int e = tid; e < G; e += kTopkThreads) {
            const u64 c = ((u64)ord_f32(sc[e]) << 32) | (uint32_t)(~(uint32_t)e);
            if (c < prev && c > best) best = c;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const u64 o = ((u64)(uint32_t)__shfl_xor((int)(best >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, off);
            best = o > best ? o : best;
        }
        if (lane == 0) s_part[wave] = best;
        __syncthreads();
        if (tid == 0) {
            u64 b = 0;
#pragma unroll
            for (int w = 0; w < kTopkThreads / 64; ++w) b = s_part[w] > b ? s_part[w] : b;
            s_best = b;
            const size_t o = (size_t)qi * k + r;
            if (b) { const uint32_t idx = ~(uint32_t)b; out_idx[o] = index_base + (long long)idx; out_score[o] = sc[idx]; }
            else { out_idx[o] = -1; out_score[o] = -INFINITY; }              // k > G
        }
        __syncthreads();
        prev = s_best;                                                     // 0 once the scores are exhausted: nothing is below it
    }
}

hipError_t launch_topk(const float* scores, int n_q, int G, int k, long long index_base, long long* out_idx, float* out_score, hipStream_t stream)
{
    if (n_q <= 0 || k <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_topk, dim3(n_q), dim3(kTopkThreads), 0, stream, scores, G, k, index_base, out_idx, out_score);
    return hipGetLastError();
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
