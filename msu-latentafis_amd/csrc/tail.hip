// tail.hip — everything after the ADC stage of a (latent, rolled) pair:
//   texture path : S7 top-200 rows, S8b distance graph (table look-up, 3 power iterations), S9 angle graph
//   minutiae path: S1 descriptor similarity, S2 normalisation, S3 top-120, S8a distance graph (sqrtf, 5 iterations), S9
//   S10 fusion.
// Reference: matching/matcher.cpp:420-516 (One2One_minutiae_matching), :723-783 (tail of One2One_texture_matching),
// :1225-1348 (LSS_R_Fast2_Dist_lookup), :1350-1469 (LSS_R_Fast2_Dist_eigen), :1471-1647 (LSS_R_Fast2, adjust_angle),
// :376-417 + :188 (template selection and fusion).
//
// One 256-thread workgroup per (pair, scorer) task, workgroups persistent over a strided task list.  Every float
// reduction keeps the reference's sequential order (index ascending, product and sum rounded separately; compiled
// with -ffp-contract=off); where the reference's order is Eigen's (unpinned) the canonical order of
// oracle/afis_oracle.cpp is used.  Equal sort keys are ordered by ascending index.
#include "afis_device.h"

namespace afis {

constexpr int kTailThreads = 256;
constexpr int kTailWaves = kTailThreads / 64;
#define AFIS_PI 3.1415926   /* matching/include.h:22 — a double literal; comparisons against it are in double */

// ---- small helpers ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_f32(float v)
{
    v = v + 0.0f;                                   // -0 -> +0 so that equal floats get equal keys
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct CountSlots { int slot[2][kTailWaves]; int parity; };

// Workgroup-wide count of a predicate; `local` is this thread's count.  One barrier per call.
__device__ __forceinline__ int wg_count(int local, int* s_slots /*[2][kTailWaves]*/, int& parity)
{
    // wave sum by ballot-free butterfly
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_slots[parity * kTailWaves + wave] = local;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < kTailWaves; ++w) tot += s_slots[parity * kTailWaves + w];
    parity ^= 1;
    return tot;
}

// Bitonic sort of P (power of two) 64-bit keys in LDS, DESCENDING.  Keys are (ord(value) << 32) | ~index, so the
// result is value-descending with ascending index on equal values.
__device__ void bitonic_desc(unsigned long long* keys, int P)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += kTailThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool desc_block = (i & k) == 0;
                    if (desc_block ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// Select the K largest of n keys (key(e) -> uint32, larger = better; equal keys: smaller e first) into
// list[0..K) as composite 64-bit keys (unsorted).  n >= K >= 1.  Workgroup-wide.
template <class KeyFn>
__device__ void select_topk(int n, int K, KeyFn key, unsigned long long* list, int* s_slots, int& parity, int* s_counter)
{
    // 1. T = K-th largest key: the largest t with count(key >= t) >= K, built bit by bit.
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        int c = 0;
        for (int e = threadIdx.x; e < n; e += kTailThreads) c += key(e) >= cand;
        if (wg_count(c, s_slots, parity) >= K) T = cand;
    }
    int cg = 0, ce = 0;
    for (int e = threadIdx.x; e < n; e += kTailThreads) { const uint32_t k = key(e); cg += k > T; ce += k == T; }
    const int n_gt = wg_count(cg, s_slots, parity);
    const int n_eq = wg_count(ce, s_slots, parity);
    const int need = K - n_gt;                         // >= 1
    // 2. among the keys equal to T keep the `need` smallest indices: B = largest bound with count(e < B) < need
    uint32_t B = 0xffffffffu;
    if (n_eq != need) {
        B = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t cand = B | (1u << bit);
            int c = 0;
            for (int e = threadIdx.x; e < n; e += kTailThreads) c += (key(e) == T) && ((uint32_t)e < cand);
            if (wg_count(c, s_slots, parity) < need) B = cand;
        }
    }
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += kTailThreads) {
        const uint32_t k = key(e);
        if (k > T || (k == T && (uint32_t)e <= B)) {
            const int pos = atomicAdd(s_counter, 1);
            list[pos] = ((unsigned long long)k << 32) | (uint32_t)(~(uint32_t)e);
        }
    }
    __syncthreads();
}

// ---- correspondence list shared by the graph stages ------------------------------------------------------------------
template <int NMAX>
struct Cands {
    float sim[NMAX];
    int li[NMAX], ri[NMAX];
    int lx[NMAX], ly[NMAX], rx[NMAX], ry[NMAX];
    float lo[NMAX], ro[NMAX];
};

template <int NMAX>
struct GraphSmem {
    Cands<NMAX> c;
    float H[NMAX * (NMAX - 1) / 2];        // strict upper triangle of the symmetric compatibility matrix
    uint32_t hb[NMAX][(NMAX + 31) / 32];   // boolean angle-compatibility matrix, bit rows
    float b[NMAX], cc[NMAX];
    unsigned long long keys[256];
    int sel[NMAX];
    int nsel;
    int slots[2 * kTailWaves];
    int counter;
};

__device__ __forceinline__ int tri(int i, int j, int num) { return i * (2 * num - i - 1) / 2 + (j - i - 1); }   // i < j

// Greedy selection, matcher.cpp:1304-1344 / :1425-1465 / :1593-1633: walk the candidates by descending S; stop at
// S < thr; skip a candidate whose latent or rolled point is already used or that is incompatible with ANY accepted one.
// Run by wave 0; accepted indices go to sm.sel[0..nsel).
template <int NMAX, class Compat>
__device__ void greedy(GraphSmem<NMAX>& sm, int num, double thr, Compat compatible)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        constexpr int U = (NMAX + 63) / 64;
        int s_ind[U], s_li[U], s_ri[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { s_ind[u] = -1; s_li[u] = -1; s_ri[u] = -1; }
        int nsel = 0;
        for (int r = 0; r < num; ++r) {
            const int ind = (int)(~(uint32_t)sm.keys[r]);
            const float s = sm.b[ind];
            if ((double)s < thr) break;
            const int li = sm.c.li[ind], ri = sm.c.ri[ind];
            bool bad = false;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (lane + 64 * u < nsel)
                    bad |= (s_li[u] == li) | (s_ri[u] == ri) | !compatible(ind, s_ind[u]);
            if (__any(bad)) continue;
            const int slot = nsel >> 6, ln = nsel & 63;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u == slot && lane == ln) { s_ind[u] = ind; s_li[u] = li; s_ri[u] = ri; }
            if (lane == 0) sm.sel[nsel] = ind;
            ++nsel;
        }
        if (lane == 0) sm.nsel = nsel;
    }
    __syncthreads();
}

// keep only the accepted correspondences, in acceptance order
template <int NMAX>
__device__ int compact(GraphSmem<NMAX>& sm)
{
    const int n = sm.nsel;
    const int t = threadIdx.x;
    float sim = 0, lo = 0, ro = 0; int li = 0, ri = 0, lx = 0, ly = 0, rx = 0, ry = 0;
    if (t < n) {
        const int s = sm.sel[t];
        sim = sm.c.sim[s]; li = sm.c.li[s]; ri = sm.c.ri[s]; lx = sm.c.lx[s]; ly = sm.c.ly[s]; rx = sm.c.rx[s]; ry = sm.c.ry[s];
        lo = sm.c.lo[s]; ro = sm.c.ro[s];
    }
    __syncthreads();
    if (t < n) {
        sm.c.sim[t] = sim; sm.c.li[t] = li; sm.c.ri[t] = ri; sm.c.lx[t] = lx; sm.c.ly[t] = ly; sm.c.rx[t] = rx; sm.c.ry[t] = ry;
        sm.c.lo[t] = lo; sm.c.ro[t] = ro;
    }
    __syncthreads();
    return n;
}

__device__ __forceinline__ void sort_scores(float* S, unsigned long long* keys, int num)
{
    const int P = next_pow2(num);
    for (int i = threadIdx.x; i < P; i += kTailThreads)
        keys[i] = i < num ? (((unsigned long long)ord_f32(S[i]) << 32) | (uint32_t)(~(uint32_t)i)) : 0ull;
    __syncthreads();
    bitonic_desc(keys, P);
}

// S8a (LOOKUP = false, 5 iterations) / S8b (LOOKUP = true, 3 iterations).
template <int NMAX, bool LOOKUP, int ITERS>
__device__ int dist_filter(GraphSmem<NMAX>& sm, int num, const float* s_table)
{
    // compatibility matrix, matcher.cpp:1237-1275 / :1363-1397
    for (int idx = threadIdx.x; idx < num * num; idx += kTailThreads) {
        const int i = idx / num, j = idx - i * num;
        if (i >= j) continue;
        float h = 0.0f;
        float d1, d2; bool ok = true;
        if (LOOKUP) {
            const int dx1 = abs(sm.c.lx[i] - sm.c.lx[j]), dx2 = abs(sm.c.rx[i] - sm.c.rx[j]);
            const int dy1 = abs(sm.c.ly[i] - sm.c.ly[j]), dy2 = abs(sm.c.ry[i] - sm.c.ry[j]);
            ok = !((dx1 >= kDistN) | (dx2 >= kDistN) | (dy1 >= kDistN) | (dy2 >= kDistN));      // :1257
            d1 = ok ? s_table[dx1 * kDistN + dy1] : 0.f;
            d2 = ok ? s_table[dx2 * kDistN + dy2] : 0.f;
        } else {
            const float dx1 = (float)(sm.c.lx[i] - sm.c.lx[j]), dx2 = (float)(sm.c.rx[i] - sm.c.rx[j]);
            const float dy1 = (float)(sm.c.ly[i] - sm.c.ly[j]), dy2 = (float)(sm.c.ry[i] - sm.c.ry[j]);
            const float a = dx1 * dx1, b = dy1 * dy1, c = dx2 * dx2, d = dy2 * dy2;
            d1 = __fsqrt_rn(a + b);                                                               // :1380-1384, correctly rounded
            d2 = __fsqrt_rn(c + d);
        }
        const float dist = fabsf(d1 - d2);
        if (ok && !(dist > 30.0f)) {
            // (30-dist)/(25.0): float numerator, double divide, float store (:1268/:1389).  A correctly rounded
            // fp32 divide gives the same float (double rounding through 53 bits is innocuous for a quotient of two
            // 24-bit values); HIP's default fp32 '/' is correctly rounded.
            h = __fdiv_rn(30.0f - dist, 25.0f);
            if (h > 1.0f) h = 1.0f; else if (h < 0.0f) h = 0.0f;
        }
        sm.H[tri(i, j, num)] = h;
    }
    for (int i = threadIdx.x; i < num; i += kTailThreads) sm.b[i] = sm.c.sim[i];
    __syncthreads();
    // power iteration, :1284-1289 / :1406-1411 (canonical order: k ascending, unfused; see oracle)
    for (int it = 0; it < ITERS; ++it) {
        const int j = threadIdx.x;
        if (j < num) {
            float acc = 0.0f;
            for (int k = 0; k < num; ++k) {
                const float h = (k == j) ? 0.0f : (k < j ? sm.H[tri(k, j, num)] : sm.H[tri(j, k, num)]);
                const float p = h * sm.b[k];
                acc += p;
            }
            sm.cc[j] = acc;
        }
        __syncthreads();
        float sum = 0.0f;
        for (int k = 0; k < num; ++k) sum += sm.cc[k];
        const float scale = (float)(1.0 / ((double)sum + 0.00001));
        if (j < num) sm.b[j] = sm.cc[j] * scale;
        __syncthreads();
    }
    sort_scores(sm.b, sm.keys, num);
    greedy<NMAX>(sm, num, 0.0001, [&sm, num](int a, int o) {
        const float h = a < o ? sm.H[tri(a, o, num)] : sm.H[tri(o, a, num)];
        return !((double)h < 0.00001);
    });
    return compact(sm);
}

__device__ __forceinline__ float adjust_angle(float angle)            // matcher.cpp:1638-1647
{
    if ((double)angle > AFIS_PI) angle = (float)((double)angle - 2 * AFIS_PI);
    else if ((double)angle < -AFIS_PI) angle = (float)((double)angle + 2 * AFIS_PI);
    return angle;
}
__device__ __forceinline__ float fold_pi(float d)                      // "if(angle_diff>PI) angle_diff = 2*PI - angle_diff"
{
    if ((double)d > AFIS_PI) d = (float)(2 * AFIS_PI - (double)d);
    return d;
}
// atan2f of the reference (glibc) replaced by a double-precision atan2 rounded to float: equal to a correctly
// rounded atan2f except for results within 1e-16 relative of a rounding boundary.  The value only feeds threshold tests.
__device__ __forceinline__ float atan2_f32(float y, float x) { return (float)atan2((double)y, (double)x); }

// S9, matcher.cpp:1471-1636
template <int NMAX>
__device__ int angle_filter(GraphSmem<NMAX>& sm, int num)
{
    constexpr int W = (NMAX + 31) / 32;
    for (int i = threadIdx.x; i < num * W; i += kTailThreads) sm.hb[i / W][i % W] = 0u;
    __syncthreads();
    for (int idx = threadIdx.x; idx < num * num; idx += kTailThreads) {
        const int i = idx / num, j = idx - i * num;
        if (i >= j) continue;
        const float lo1 = sm.c.lo[i], lo2 = sm.c.lo[j], ro1 = sm.c.ro[i], ro2 = sm.c.ro[j];
        float angle_1 = adjust_angle(lo1 - lo2);
        float angle_2 = adjust_angle(ro1 - ro2);
        float angle_diff = fold_pi(fabsf(angle_1 - angle_2));
        if ((double)angle_diff > AFIS_PI / 4.) continue;
        const float dx_1 = (float)(sm.c.lx[i] - sm.c.lx[j]), dy_1 = (float)(sm.c.ly[i] - sm.c.ly[j]);
        const float line_angle_1 = -atan2_f32(dy_1, dx_1);
        angle_1 = adjust_angle(lo1 - line_angle_1);
        const float dx_2 = (float)(sm.c.rx[i] - sm.c.rx[j]), dy_2 = (float)(sm.c.ry[i] - sm.c.ry[j]);
        const float line_angle_2 = -atan2_f32(dy_2, dx_2);
        angle_2 = adjust_angle(ro1 - line_angle_2);
        angle_diff = fold_pi(fabsf(angle_1 - angle_2));
        if ((double)angle_diff > AFIS_PI / 6.) continue;
        angle_1 = adjust_angle(lo2 - line_angle_1);
        angle_2 = adjust_angle(ro2 - line_angle_2);
        angle_diff = fold_pi(fabsf(angle_1 - angle_2));
        if ((double)angle_diff > AFIS_PI / 6.) continue;
        atomicOr(&sm.hb[i][j >> 5], 1u << (j & 31));
        atomicOr(&sm.hb[j][i >> 5], 1u << (i & 31));
    }
    const float s0 = (float)(1.0 / num);                                // :1558
    for (int i = threadIdx.x; i < num; i += kTailThreads) sm.b[i] = s0;
    __syncthreads();
    for (int it = 0; it < 5; ++it) {                                    // :1563-1581
        const int j = threadIdx.x;
        if (j < num) {
            float s1 = 0.0f;
            for (int w = 0; w < (num + 31) / 32; ++w) {
                uint32_t bits = sm.hb[j][w];
                while (bits) { const int k = w * 32 + __ffs(bits) - 1; bits &= bits - 1; s1 += sm.b[k]; }
            }
            sm.cc[j] = s1;
        }
        __syncthreads();
        float sum = 0.0f;
        for (int k = 0; k < num; ++k) sum += sm.cc[k];
        sum = (float)(1.0 / ((double)sum + 0.00001));
        if (j < num) sm.b[j] = sm.cc[j] * sum;
        __syncthreads();
    }
    sort_scores(sm.b, sm.keys, num);
    greedy<NMAX>(sm, num, 0.001, [&sm](int a, int o) { return (sm.hb[a][o >> 5] >> (o & 31)) & 1u; });
    return compact(sm);
}

template <int NMAX>
__device__ __forceinline__ float sum_sims(const GraphSmem<NMAX>& sm, int n)   // :508-514 / :775-781
{
    float score = 0.0f;
    for (int i = 0; i < n; ++i) score += sm.c.sim[i];
    return score;
}

// =====================================================================================================================
// texture tail
// =====================================================================================================================
struct TexSmem {
    GraphSmem<kTopTex> g;
    float table[kDistN * kDistN];
    float val[kTexMax];
    int arg[kTexMax];
    uint32_t key[kTexMax];
};

__global__ __launch_bounds__(kTailThreads) void k_texture_tail(QueryDev q, GalleryDev g, const float* __restrict__ table_dist,
                                                               const float* __restrict__ rm_val, const int32_t* __restrict__ rm_arg,
                                                               float* __restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TexSmem& sm = *reinterpret_cast<TexSmem*>(smem_raw);
    for (int i = threadIdx.x; i < kDistN * kDistN; i += kTailThreads) sm.table[i] = table_dist[i];
    __syncthreads();
    int parity = 0;
    const long long n_tasks = (long long)q.nq * g.G;
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int qi = (int)(task / g.G), gi = (int)(task - (long long)qi * g.G);
        const int l0 = q.lt_off[qi], n_lt = q.lt_off[qi + 1] - l0;
        const int r0 = g.tex_off[gi], n_rt = g.tex_off[gi + 1] - r0;
        float* out = parts + (size_t)task * 4 + 3;
        if (n_lt <= 0 || n_rt <= 0) { if (threadIdx.x == 0) *out = 0.0f; continue; }   // matcher.cpp:411: scorer not called
        const size_t o = (size_t)task * q.lt_pad;
        for (int i = threadIdx.x; i < n_lt; i += kTailThreads) {
            const float v = rm_val[o + i];
            sm.val[i] = v; sm.arg[i] = rm_arg[o + i]; sm.key[i] = ord_f32(v);
        }
        __syncthreads();
        int num;
        if (n_lt > kTopTex) {                                            // :736-747
            select_topk(n_lt, kTopTex, [&sm](int e) { return sm.key[e]; }, sm.g.keys, sm.g.slots, parity, &sm.g.counter);
            for (int i = kTopTex + threadIdx.x; i < 256; i += kTailThreads) sm.g.keys[i] = 0ull;
            __syncthreads();
            bitonic_desc(sm.g.keys, 256);
            num = kTopTex;
            if (threadIdx.x < num) {
                const int e = (int)(~(uint32_t)sm.g.keys[threadIdx.x]);
                sm.g.c.sim[threadIdx.x] = sm.val[e]; sm.g.c.li[threadIdx.x] = e; sm.g.c.ri[threadIdx.x] = sm.arg[e];
            }
        } else {                                                         // :748-749 rows stay in index order
            num = n_lt;
            if (threadIdx.x < num) {
                sm.g.c.sim[threadIdx.x] = sm.val[threadIdx.x]; sm.g.c.li[threadIdx.x] = threadIdx.x; sm.g.c.ri[threadIdx.x] = sm.arg[threadIdx.x];
            }
        }
        __syncthreads();
        if (threadIdx.x < num) {
            const int t = threadIdx.x;
            const short2 lp = q.lt_xy[l0 + sm.g.c.li[t]], rp = g.tex_xy[r0 + sm.g.c.ri[t]];
            sm.g.c.lx[t] = lp.x; sm.g.c.ly[t] = lp.y; sm.g.c.rx[t] = rp.x; sm.g.c.ry[t] = rp.y;
            sm.g.c.lo[t] = q.lt_ori[l0 + sm.g.c.li[t]]; sm.g.c.ro[t] = g.tex_ori[r0 + sm.g.c.ri[t]];
        }
        __syncthreads();
        num = dist_filter<kTopTex, true, 3>(sm.g, num, sm.table);        // :759
        num = angle_filter<kTopTex>(sm.g, num);                          // :767
        if (threadIdx.x == 0) *out = sum_sims(sm.g, num);
        __syncthreads();
    }
}

hipError_t launch_texture_tail(const QueryDev& q, const GalleryDev& g, const float* table_dist,
                               const float* rm_val, const int32_t* rm_arg, float* parts, hipStream_t stream)
{
    const long long n_tasks = (long long)q.nq * g.G;
    if (n_tasks <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_texture_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TexSmem));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (int)(n_tasks < 2048 ? n_tasks : 2048);
    hipLaunchKernelGGL(k_texture_tail, dim3(grid), dim3(kTailThreads), sizeof(TexSmem), stream, q, g, table_dist, rm_val, rm_arg, parts);
    return hipGetLastError();
}

// =====================================================================================================================
// minutiae scorer
// =====================================================================================================================
constexpr int kGemmTile = 64;
constexpr int kGemmLd = 100;          // padded row stride (floats): 16-byte aligned rows, conflict-free b128 column walks
constexpr int kMinuMaxPts = 2000;     // Max_Nrof_Minutiae, matcher.cpp:788

struct MinuSmem {
    union {
        struct { float A[kGemmTile * kGemmLd]; float B[kGemmTile * kGemmLd]; } t;    // 51.2 KB, GEMM phase
        GraphSmem<kTopMinu> g;                                                        // graph phase
    } u;
    float rowsum[kMinuMaxPts];
    float colsum[kMinuMaxPts];
};

__global__ __launch_bounds__(kTailThreads) void k_minutiae(QueryDev q, GalleryDev g, float* __restrict__ scratch, size_t scratch_per_wg,
                                                           float* __restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MinuSmem& sm = *reinterpret_cast<MinuSmem*>(smem_raw);
    float* simi = scratch + (size_t)blockIdx.x * scratch_per_wg;
    int parity = 0;
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    const int tid = threadIdx.x;
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        // task order: gallery template fastest, then selected template, then query
        const int gi = (int)(task % g.G);
        const int qs = (int)(task / g.G);                    // qi*3 + s
        const int qi = qs / 3, s = qs - qi * 3;
        const int l0 = q.lm_off[qs], nL = q.lm_off[qs + 1] - l0;
        const int r0 = g.minu_off[gi], nR = g.minu_off[gi + 1] - r0;
        float* out = parts + ((size_t)qi * g.G + gi) * 4 + s;
        if (nL <= 0 || nR <= 0) { if (tid == 0) *out = 0.0f; continue; }     // matcher.cpp:400-404
        const int n = nL * nR;
        uint32_t* keys = reinterpret_cast<uint32_t*>(simi + (scratch_per_wg >> 1));

        // ---- S1: simi = max(0, A * B^T), canonical order = fmaf chain, k ascending (matcher.cpp:440-452) ----
        for (int it = 0; it < nL; it += kGemmTile) {
            for (int jt = 0; jt < nR; jt += kGemmTile) {
                __syncthreads();
                for (int e = tid; e < kGemmTile * (kDes / 4); e += kTailThreads) {
                    const int r = e / (kDes / 4), k4 = e - r * (kDes / 4);
                    float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
                    if (it + r < nL) a = *reinterpret_cast<const float4*>(q.lm_des + (size_t)(l0 + it + r) * kDes + k4 * 4);
                    if (jt + r < nR) b = *reinterpret_cast<const float4*>(g.minu_des + (size_t)(r0 + jt + r) * kDes + k4 * 4);
                    *reinterpret_cast<float4*>(&sm.u.t.A[r * kGemmLd + k4 * 4]) = a;
                    *reinterpret_cast<float4*>(&sm.u.t.B[r * kGemmLd + k4 * 4]) = b;
                }
                __syncthreads();
                const int ty = tid >> 4, tx = tid & 15;              // rows ty + 16*r, cols tx + 16*c
                float acc[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
                for (int k4 = 0; k4 < kDes / 4; ++k4) {
                    float4 a[4], b[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const float4*>(&sm.u.t.A[(ty + 16 * r) * kGemmLd + k4 * 4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(&sm.u.t.B[(tx + 16 * c) * kGemmLd + k4 * 4]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = acc[r][c];
                            v = fmaf(a[r].x, b[c].x, v); v = fmaf(a[r].y, b[c].y, v);
                            v = fmaf(a[r].z, b[c].z, v); v = fmaf(a[r].w, b[c].w, v);
                            acc[r][c] = v;
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = it + ty + 16 * r, j = jt + tx + 16 * c;
                        if (i < nL && j < nR) { float v = acc[r][c]; if (v < 0) v = 0; simi[(size_t)i * nR + j] = v; }
                    }
            }
        }
        __syncthreads();
        // ---- S2: column sums (rolled) / row sums (latent), index ascending (:455-456) ----
        for (int j = tid; j < nR; j += kTailThreads) { float sacc = 0.f; for (int i = 0; i < nL; ++i) sacc += simi[(size_t)i * nR + j]; sm.colsum[j] = sacc; }
        for (int i = tid; i < nL; i += kTailThreads) { float sacc = 0.f; for (int j = 0; j < nR; ++j) sacc += simi[(size_t)i * nR + j]; sm.rowsum[i] = sacc; }
        __syncthreads();
        for (int e = tid; e < n; e += kTailThreads) {                   // :461-470
            const int i = e / nR, j = e - i * nR;
            const float sv = simi[e];
            float f = sm.rowsum[i] + sm.colsum[j];
            f = f - sv;
            const float norm = (float)((double)sv / ((double)f + 0.000001));
            keys[e] = ord_f32(norm);
        }
        __syncthreads();
        // ---- S3: top-120 by normalised similarity (:473-488) ----
        GraphSmem<kTopMinu>& gs = sm.u.g;
        const int topN = n < kTopMinu ? n : kTopMinu;
        select_topk(n, topN, [keys](int e) { return keys[e]; }, gs.keys, gs.slots, parity, &gs.counter);
        const int P = next_pow2(topN);
        for (int i = topN + tid; i < P; i += kTailThreads) gs.keys[i] = 0ull;
        __syncthreads();
        bitonic_desc(gs.keys, P);
        if (tid < topN) {
            const int e = (int)(~(uint32_t)gs.keys[tid]);
            const int i1 = e / nR, i2 = e - i1 * nR;
            gs.c.sim[tid] = simi[e]; gs.c.li[tid] = i1; gs.c.ri[tid] = i2;
            const short2 lp = q.lm_xy[l0 + i1], rp = g.minu_xy[r0 + i2];
            gs.c.lx[tid] = lp.x; gs.c.ly[tid] = lp.y; gs.c.rx[tid] = rp.x; gs.c.ry[tid] = rp.y;
            gs.c.lo[tid] = q.lm_ori[l0 + i1]; gs.c.ro[tid] = g.minu_ori[r0 + i2];
        }
        __syncthreads();
        int num = dist_filter<kTopMinu, false, 5>(gs, topN, nullptr);    // :492
        num = angle_filter<kTopMinu>(gs, num);                            // :495
        if (tid == 0) *out = sum_sims(gs, num);
        __syncthreads();
    }
}

hipError_t launch_minutiae(const QueryDev& q, const GalleryDev& g, float* scratch, size_t scratch_floats_per_wg, int n_wg,
                           float* parts, hipStream_t stream)
{
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    if (n_tasks <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_minutiae), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MinuSmem));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (int)(n_tasks < n_wg ? n_tasks : n_wg);
    hipLaunchKernelGGL(k_minutiae, dim3(grid), dim3(kTailThreads), sizeof(MinuSmem), stream, q, g, scratch, scratch_floats_per_wg, parts);
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

hipError_t launch_fuse(const QueryDev& q, const GalleryDev& g, const float* parts, float* scores, hipStream_t stream)
{
    const long long n = (long long)q.nq * g.G;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fuse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q, g, parts, scores);
    return hipGetLastError();
}

}  // namespace afis
