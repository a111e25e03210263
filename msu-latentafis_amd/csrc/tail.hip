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
//
// How the reference's serial steps are mapped onto a workgroup without changing their results:
//   * full std::sort of <= 256 keys     -> rank by counting (thread t counts keys larger than its own), one barrier;
//   * "top-N of n" (N = 200 / 120)      -> bit-by-bit search for the N-th largest key with ballot/popcount counts
//                                          (texture: every wave holds all <= 1000 keys in registers, no barrier at all),
//                                          then the N survivors are rank-sorted;
//   * greedy clique selection            -> rounds: the first still-alive candidate in rank order is accepted and every
//                                          later candidate that conflicts with it is killed in parallel; identical to the
//                                          sequential scan, but the trip count is the number of ACCEPTED candidates;
//   * float sums                         -> every thread adds the same values in the same (ascending) order.
#include "afis_device.h"

namespace afis {

constexpr int kTailThreads = 256;
constexpr int kTailWaves = kTailThreads / 64;
#define AFIS_PI 3.1415926   /* matching/include.h:22 — a double literal; comparisons against it are in double */

typedef unsigned long long u64;

// ---- small helpers ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_f32(float v)
{
    v = v + 0.0f;                                   // -0 -> +0 so that equal floats get equal keys
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ u64 make_key(float v, int idx) { return ((u64)ord_f32(v) << 32) | (uint32_t)(~(uint32_t)idx); }
__device__ __forceinline__ int wave_popc(bool p) { return __popcll(__ballot(p)); }
__device__ __forceinline__ int lane_prefix(u64 mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0)); }

// Workgroup-wide sum of per-thread counts.  One barrier per call (slots alternate by parity).
__device__ __forceinline__ int wg_count(int local, int* s_slots /*[2][kTailWaves]*/, int& parity)
{
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

// ---- correspondence list shared by the graph stages ------------------------------------------------------------------
template <int NMAX>
struct Cands {
    float sim[NMAX];
    short li[NMAX], ri[NMAX];
    short lx[NMAX], ly[NMAX], rx[NMAX], ry[NMAX];
    float lo[NMAX], ro[NMAX];
};

template <int NMAX>
struct __attribute__((aligned(16))) GraphSmem {
    float b[(NMAX + 3) / 4 * 4];           // 16-byte aligned: read as float4 broadcasts
    float cc[(NMAX + 3) / 4 * 4];
    Cands<NMAX> c;
    float H[NMAX * (NMAX - 1) / 2];        // strict upper triangle of the symmetric compatibility matrix
    uint32_t hb[NMAX][(NMAX + 31) / 32];   // boolean angle-compatibility matrix, bit rows
    u64 keys[256];
    short order[256];                      // rank -> candidate index
    short sel[NMAX];
    int nsel;
    int slots[2 * kTailWaves];
    int counter;
};

__device__ __forceinline__ int tri_off(int i, int num) { return i * (2 * num - i - 1) / 2; }
__device__ __forceinline__ int tri(int i, int j, int num) { return tri_off(i, num) + (j - i - 1); }   // i < j
// inverse of tri(): pair number p -> (i, j), i < j
__device__ __forceinline__ void tri_inv(int p, int num, int& i, int& j)
{
    const float fn = (float)(2 * num - 1);
    int r = (int)((fn - __fsqrt_rn(fn * fn - 8.0f * (float)p)) * 0.5f);    // within +-1 of the row for num <= 256
    r = max(0, min(r, num - 2));
    r -= (tri_off(r, num) > p);                    // branch-free fix-up
    r += (tri_off(r + 1, num) <= p);
    r -= (tri_off(r, num) > p);
    i = r; j = p - tri_off(r, num) + r + 1;
}

// Rank (0 = largest) of this thread's key among keys[0..n); keys are unique.  Caller syncs before (keys written) and after.
__device__ __forceinline__ int rank_of(const u64* keys, int n, u64 mine)
{
    int r = 0;
#pragma unroll 8
    for (int k = 0; k < n; ++k) r += keys[k] > mine;
    return r;
}

// sort the candidates by score (descending, ties by index): order[rank] = index.
template <int NMAX>
__device__ __forceinline__ void sort_scores(GraphSmem<NMAX>& sm, int num)
{
    const int t = threadIdx.x;
    u64 mine = 0;
    if (t < num) { mine = make_key(sm.b[t], t); sm.keys[t] = mine; }
    __syncthreads();
    if (t < num) sm.order[rank_of(sm.keys, num, mine)] = (short)t;
    __syncthreads();
}

// Greedy selection, matcher.cpp:1304-1344 / :1425-1465 / :1593-1633: walk the candidates by descending S; stop at S < thr;
// skip a candidate whose latent or rolled point is already used or that is incompatible with ANY accepted one.
// Wave 0 holds the candidates in rank order (lane l: ranks l, l+64, ...).  Each round accepts the first alive candidate and
// kills every later one that conflicts with it.  Accepted indices go to sm.sel[0..nsel) in acceptance (= rank) order.
template <int NMAX, class Compat>
__device__ void greedy(GraphSmem<NMAX>& sm, int num, double thr, Compat compatible)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        constexpr int U = (NMAX + 63) / 64;
        int idx[U], li[U], ri[U];
        bool alive[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = lane + 64 * u;
            idx[u] = 0; li[u] = -1; ri[u] = -1; alive[u] = false;
            if (p < num) {
                idx[u] = sm.order[p];
                li[u] = sm.c.li[idx[u]]; ri[u] = sm.c.ri[idx[u]];
                alive[u] = !((double)sm.b[idx[u]] < thr);          // sorted descending: everything after the first S < thr is < thr too
            }
        }
        int nsel = 0;
        for (;;) {
            int first = -1, cidx = 0, cli = 0, cri = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u64 m = __ballot(alive[u]);
                if (first < 0 && m) {
                    const int fl = __ffsll((long long)m) - 1;
                    first = 64 * u + fl;
                    cidx = __shfl(idx[u], fl); cli = __shfl(li[u], fl); cri = __shfl(ri[u], fl);
                }
            }
            if (first < 0) break;
            if (lane == 0) sm.sel[nsel] = (short)cidx;
            ++nsel;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (alive[u]) {
                    const int p = lane + 64 * u;
                    if (p == first || li[u] == cli || ri[u] == cri || !compatible(cidx, idx[u])) alive[u] = false;
                }
            }
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
    float sim = 0, lo = 0, ro = 0; short li = 0, ri = 0, lx = 0, ly = 0, rx = 0, ry = 0;
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

// sum of cc[0..num) in ascending order; cc is zero-padded to a multiple of 4 (x + 0.0f == x)
template <int NMAX>
__device__ __forceinline__ float seq_sum(const GraphSmem<NMAX>& sm, int num)
{
    float sum = 0.0f;
    const float4* c4 = reinterpret_cast<const float4*>(sm.cc);
#pragma unroll 4
    for (int k = 0; k < (num + 3) / 4; ++k) { const float4 v = c4[k]; sum += v.x; sum += v.y; sum += v.z; sum += v.w; }
    return sum;
}

// S8a (LOOKUP = false, 5 iterations) / S8b (LOOKUP = true, 3 iterations).
template <int NMAX, bool LOOKUP, int ITERS>
__device__ int dist_filter(GraphSmem<NMAX>& sm, int num, const float* s_table)
{
    const int t = threadIdx.x;
    // compatibility matrix, matcher.cpp:1237-1275 / :1363-1397
    const int n_pairs = num * (num - 1) / 2;
#pragma unroll 2
    for (int p = t; p < n_pairs; p += kTailThreads) {
        int i, j; tri_inv(p, num, i, j);
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
            // 24-bit values, 53 >= 2*24+2); __fdiv_rn is correctly rounded.
            h = __fdiv_rn(30.0f - dist, 25.0f);
            if (h > 1.0f) h = 1.0f; else if (h < 0.0f) h = 0.0f;
        }
        sm.H[p] = h;
    }
    if (t < (NMAX + 3) / 4 * 4) { sm.b[t] = t < num ? sm.c.sim[t] : 0.0f; sm.cc[t] = 0.0f; }
    __syncthreads();
    // power iteration, :1284-1289 / :1406-1411 (canonical order: k ascending, unfused; see oracle)
    for (int it = 0; it < ITERS; ++it) {
        if (t < num) {
            float acc = 0.0f;
            int a = t - 1;                               // address of H[0][t] in the triangle (k < t part walks down column t)
#pragma unroll 4
            for (int k = 0; k < t; ++k) {                // H[k][t]
                const float p = sm.H[a] * sm.b[k];
                acc += p;
                a += num - k - 2;
            }
            // k == t contributes H[t][t]*b[t] = 0 (+0.0f leaves acc unchanged)
            a = tri_off(t, num);
#pragma unroll 4
            for (int k = t + 1; k < num; ++k) {          // H[t][k], contiguous
                const float p = sm.H[a] * sm.b[k];
                acc += p;
                ++a;
            }
            sm.cc[t] = acc;
        }
        __syncthreads();
        const float sum = seq_sum(sm, num);
        const float scale = (float)(1.0 / ((double)sum + 0.00001));
        if (t < num) sm.b[t] = sm.cc[t] * scale;          // nobody reads b between the barrier above and the one below
        __syncthreads();
    }
    sort_scores(sm, num);
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
    const int t = threadIdx.x;
    for (int i = t; i < num * W; i += kTailThreads) sm.hb[i / W][i % W] = 0u;
    if (t < (NMAX + 3) / 4 * 4) { sm.b[t] = t < num ? (float)(1.0 / num) : 0.0f; sm.cc[t] = 0.0f; }   // :1558
    __syncthreads();
    const int n_pairs = num * (num - 1) / 2;
    for (int p = t; p < n_pairs; p += kTailThreads) {
        int i, j; tri_inv(p, num, i, j);
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
    __syncthreads();
    for (int it = 0; it < 5; ++it) {                                    // :1563-1581
        if (t < num) {
            float s1 = 0.0f;
            for (int w = 0; w < (num + 31) / 32; ++w) {
                uint32_t bits = sm.hb[t][w];
                while (bits) { const int k = w * 32 + __ffs(bits) - 1; bits &= bits - 1; s1 += sm.b[k]; }
            }
            sm.cc[t] = s1;
        }
        __syncthreads();
        float sum = seq_sum(sm, num);
        sum = (float)(1.0 / ((double)sum + 0.00001));
        if (t < num) sm.b[t] = sm.cc[t] * sum;
        __syncthreads();
    }
    sort_scores(sm, num);
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

// both graph stages; a list of fewer than 2 correspondences cannot survive S9 (a single node ends with S = 0)
template <int NMAX, bool LOOKUP, int ITERS>
__device__ __forceinline__ float graph_score(GraphSmem<NMAX>& sm, int num, const float* s_table)
{
    num = dist_filter<NMAX, LOOKUP, ITERS>(sm, num, s_table);
    if (num < 2) return 0.0f;
    num = angle_filter<NMAX>(sm, num);
    return sum_sims(sm, num);
}

// =====================================================================================================================
// texture tail
// =====================================================================================================================
struct TexSmem {
    GraphSmem<kTopTex> g;
    float table[kDistN * kDistN];
    float tval[kTopTex];            // staging of the selected rows before they are ordered by rank
    short te[kTopTex], targ[kTopTex];
};

constexpr int kTexRegs = (kTexMax + 63) / 64;     // 16 row maxima per lane: a wave holds all <= 1000 keys in registers

__global__ __launch_bounds__(kTailThreads) void k_texture_tail(QueryDev q, GalleryDev g, const float* __restrict__ table_dist,
                                                               const float* __restrict__ rm_val, const int32_t* __restrict__ rm_arg,
                                                               float* __restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TexSmem& sm = *reinterpret_cast<TexSmem*>(smem_raw);
    for (int i = threadIdx.x; i < kDistN * kDistN; i += kTailThreads) sm.table[i] = table_dist[i];
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63;
    const long long n_tasks = (long long)q.nq * g.G;
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int qi = (int)(task / g.G), gi = (int)(task - (long long)qi * g.G);
        const int l0 = q.lt_off[qi], n_lt = q.lt_off[qi + 1] - l0;
        const int r0 = g.tex_off[gi], n_rt = g.tex_off[gi + 1] - r0;
        float* out = parts + (size_t)task * 4 + 3;
        if (n_lt <= 0 || n_rt <= 0) { if (t == 0) *out = 0.0f; continue; }   // matcher.cpp:411: scorer not called
        const size_t o = (size_t)task * q.lt_pad;
        int num;
        if (n_lt > kTopTex) {                                            // :736-747: the 200 rows with the largest maxima
            if (t < 64) {                                                // one wave, all keys in registers, no barriers
                float v[kTexRegs]; uint32_t key[kTexRegs];
#pragma unroll
                for (int u = 0; u < kTexRegs; ++u) {
                    const int e = u * 64 + lane;
                    v[u] = e < n_lt ? rm_val[o + e] : 0.0f;
                    key[u] = e < n_lt ? ord_f32(v[u]) : 0u;              // real keys are never 0
                }
                uint32_t T = 0;                                          // K-th largest key, built bit by bit
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = T | (1u << bit);
                    int c = 0;
#pragma unroll
                    for (int u = 0; u < kTexRegs; ++u) c += wave_popc(key[u] >= cand);
                    if (c >= kTopTex) T = cand;
                }
                int n_gt = 0;
#pragma unroll
                for (int u = 0; u < kTexRegs; ++u) n_gt += wave_popc(key[u] > T);
                const int need = kTopTex - n_gt;                         // of the keys equal to T keep the lowest indices
                int base_gt = 0, base_eq = 0;
#pragma unroll
                for (int u = 0; u < kTexRegs; ++u) {                     // u ascending, lane ascending = index ascending
                    const int e = u * 64 + lane;
                    const bool gt = key[u] > T, eq = key[u] == T;
                    const u64 mg = __ballot(gt), me = __ballot(eq);
                    int pos = -1;
                    if (gt) pos = base_gt + lane_prefix(mg);
                    else if (eq) { const int r = base_eq + lane_prefix(me); if (r < need) pos = n_gt + r; }
                    if (pos >= 0) {
                        sm.g.keys[pos] = make_key(v[u], e);
                        sm.tval[pos] = v[u]; sm.te[pos] = (short)e; sm.targ[pos] = (short)rm_arg[o + e];
                    }
                    base_gt += __popcll(mg); base_eq += __popcll(me);
                }
            }
            __syncthreads();
            num = kTopTex;
            int r = 0;
            if (t < num) r = rank_of(sm.g.keys, num, sm.g.keys[t]);
            if (t < num) { sm.g.c.sim[r] = sm.tval[t]; sm.g.c.li[r] = sm.te[t]; sm.g.c.ri[r] = sm.targ[t]; }
        } else {                                                         // :748-749 rows stay in index order
            num = n_lt;
            if (t < num) { sm.g.c.sim[t] = rm_val[o + t]; sm.g.c.li[t] = (short)t; sm.g.c.ri[t] = (short)rm_arg[o + t]; }
        }
        __syncthreads();
        if (t < num) {
            const short2 lp = q.lt_xy[l0 + sm.g.c.li[t]], rp = g.tex_xy[r0 + sm.g.c.ri[t]];
            sm.g.c.lx[t] = lp.x; sm.g.c.ly[t] = lp.y; sm.g.c.rx[t] = rp.x; sm.g.c.ry[t] = rp.y;
            sm.g.c.lo[t] = q.lt_ori[l0 + sm.g.c.li[t]]; sm.g.c.ro[t] = g.tex_ori[r0 + sm.g.c.ri[t]];
        }
        __syncthreads();
        const float score = graph_score<kTopTex, true, 3>(sm.g, num, sm.table);   // :759, :767
        if (t == 0) *out = score;
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
    const int grid = (int)(n_tasks < 4096 ? n_tasks : 4096);
    hipLaunchKernelGGL(k_texture_tail, dim3(grid), dim3(kTailThreads), sizeof(TexSmem), stream, q, g, table_dist, rm_val, rm_arg, parts);
    return hipGetLastError();
}

// =====================================================================================================================
// minutiae scorer
// =====================================================================================================================
constexpr int kGemmTile = 64;
constexpr int kGemmLd = 100;          // padded row stride (floats): 16-byte aligned rows, conflict-free b128 column walks
constexpr int kMinuMaxPts = 2000;     // Max_Nrof_Minutiae, matcher.cpp:788
constexpr int kKeyRegs = 16;          // per-thread keys held in registers when nL*nR <= 256*16

struct MinuSmem {
    union {
        struct { float A[kGemmTile * kGemmLd]; float B[kGemmTile * kGemmLd]; } t;    // 51.2 KB, GEMM phase
        GraphSmem<kTopMinu> g;                                                        // graph phase
    } u;
    float rowsum[kMinuMaxPts];
    float colsum[kMinuMaxPts];
    int te[kTopMinu];
};

// Select the K largest of n keys (larger = better; equal keys: smaller index first) into list[0..K) as composite keys
// (unsorted), element index into elist.  Keys come from `key(e)`; the caller decides where they live.
template <class KeyFn>
__device__ void select_topk(int n, int K, KeyFn key, u64* list, int* elist, int* s_slots, int& parity, int* s_counter)
{
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
    uint32_t B = 0xffffffffu;                          // keep the keys equal to T whose index is <= B
    if (n_eq != need) {
        B = 0;
        for (int bit = 30; bit >= 0; --bit) {          // B = largest bound with count(key == T && e < B) < need
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
            list[pos] = ((u64)k << 32) | (uint32_t)(~(uint32_t)e);
            elist[pos] = e;
        }
    }
    __syncthreads();
}

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
        uint32_t* gkeys = reinterpret_cast<uint32_t*>(simi + (scratch_per_wg >> 1));

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
        // ---- S3: top-120 by normalised similarity (:461-488) ----
        GraphSmem<kTopMinu>& gs = sm.u.g;
        const int topN = n < kTopMinu ? n : kTopMinu;
        auto norm_key = [&](int e) {
            const int i = e / nR, j = e - i * nR;
            const float sv = simi[e];
            float f = sm.rowsum[i] + sm.colsum[j];
            f = f - sv;
            return ord_f32((float)((double)sv / ((double)f + 0.000001)));                        // :467
        };
        if (n <= kTailThreads * kKeyRegs) {                              // keys in registers
            uint32_t rk[kKeyRegs];
#pragma unroll
            for (int u = 0; u < kKeyRegs; ++u) { const int e = tid + u * kTailThreads; rk[u] = e < n ? norm_key(e) : 0u; }
            // (same search as select_topk, with the register array; written out because a lambda cannot index registers dynamically)
            uint32_t T = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = T | (1u << bit);
                int c = 0;
#pragma unroll
                for (int u = 0; u < kKeyRegs; ++u) c += rk[u] >= cand;
                if (wg_count(c, gs.slots, parity) >= topN) T = cand;
            }
            int cg = 0, ce = 0;
#pragma unroll
            for (int u = 0; u < kKeyRegs; ++u) { cg += rk[u] > T; ce += (rk[u] == T) && (tid + u * kTailThreads < n); }
            const int n_gt = wg_count(cg, gs.slots, parity);
            const int n_eq = wg_count(ce, gs.slots, parity);
            const int need = topN - n_gt;
            uint32_t B = 0xffffffffu;
            if (n_eq != need) {
                B = 0;
                for (int bit = 30; bit >= 0; --bit) {
                    const uint32_t cand = B | (1u << bit);
                    int c = 0;
#pragma unroll
                    for (int u = 0; u < kKeyRegs; ++u) { const uint32_t e = tid + u * kTailThreads; c += (rk[u] == T) && (e < (uint32_t)n) && (e < cand); }
                    if (wg_count(c, gs.slots, parity) < need) B = cand;
                }
            }
            if (tid == 0) gs.counter = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kKeyRegs; ++u) {
                const int e = tid + u * kTailThreads;
                if (e < n && (rk[u] > T || (rk[u] == T && (uint32_t)e <= B))) {
                    const int pos = atomicAdd(&gs.counter, 1);
                    gs.keys[pos] = ((u64)rk[u] << 32) | (uint32_t)(~(uint32_t)e);
                    sm.te[pos] = e;
                }
            }
            __syncthreads();
        } else {                                                         // large templates: keys in the global scratch
            for (int e = tid; e < n; e += kTailThreads) gkeys[e] = norm_key(e);
            __syncthreads();
            select_topk(n, topN, [gkeys](int e) { return gkeys[e]; }, gs.keys, sm.te, gs.slots, parity, &gs.counter);
        }
        if (tid < topN) {
            const int r = rank_of(gs.keys, topN, gs.keys[tid]);
            const int e = sm.te[tid];
            const int i1 = e / nR, i2 = e - i1 * nR;
            gs.c.sim[r] = simi[e]; gs.c.li[r] = (short)i1; gs.c.ri[r] = (short)i2;
            const short2 lp = q.lm_xy[l0 + i1], rp = g.minu_xy[r0 + i2];
            gs.c.lx[r] = lp.x; gs.c.ly[r] = lp.y; gs.c.rx[r] = rp.x; gs.c.ry[r] = rp.y;
            gs.c.lo[r] = q.lm_ori[l0 + i1]; gs.c.ro[r] = g.minu_ori[r0 + i2];
        }
        __syncthreads();
        const float score = graph_score<kTopMinu, false, 5>(gs, topN, nullptr);   // :492, :495
        if (tid == 0) *out = score;
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
