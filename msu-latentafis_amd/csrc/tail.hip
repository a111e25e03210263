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

// Optional in-kernel phase timers (make PHASE_TIMING=1): thread 0 of every workgroup accumulates s_memtime deltas per phase.
#ifdef AFIS_PHASE_TIMING
__device__ u64 g_phase_cycles[32];
#define PHASE_INIT() u64 ph_t0 = __builtin_readcyclecounter()
#define PHASE(i) do { if (threadIdx.x == 0) { const u64 ph_t1 = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], ph_t1 - ph_t0); ph_t0 = ph_t1; } } while (0)
#else
#define PHASE_INIT() do {} while (0)
#define PHASE(i) do {} while (0)
#endif

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

// ---- correspondence list and scratch shared by the graph stages ------------------------------------------------------
// The compatibility matrix H of the distance graph (200x200 fp32 would be 160 KB) is never stored: LDS keeps only the
// bitmask of its NON-ZERO entries (5.6 KB); values are recomputed on demand from the packed point coordinates and cached
// per row in a small stash for the later power iterations.  H*b only ever adds h*b[k] terms, and a zero entry adds
// +0.0f, so skipping the zero entries is exact.  This keeps a task at ~45 KB of LDS (3 workgroups per CU).
template <int NMAX, int CACHE>
struct __attribute__((aligned(16))) GraphSmem {
    static constexpr int W = (NMAX + 31) / 32;
    static constexpr int N4 = (NMAX + 3) / 4 * 4;
    float b[N4];                           // 16-byte aligned: read as float4 broadcasts
    float cc[N4];
    float sim[NMAX];
    short li[NMAX], ri[NMAX];
    int2 xy[NMAX];                         // .x = lx | ly << 16 (latent point), .y = rx | ry << 16 (rolled point)
    float lo[NMAX], ro[NMAX];
    uint32_t hb[NMAX][W];                  // bit rows: non-zero pattern of H (distance stage), then the boolean H of the angle stage
    float stash[CACHE * NMAX];             // [n][t]: value of the n-th non-zero of row t
    u64 keys[256];
    short order[256];                      // rank -> candidate index
    short sel[NMAX];
    int nsel;
    int slots[2 * kTailWaves];
    int counter;
};

__device__ __forceinline__ int2 pack_xy(int lx, int ly, int rx, int ry) { return make_int2((lx & 0xffff) | (ly << 16), (rx & 0xffff) | (ry << 16)); }
struct Pt { int lx, ly, rx, ry; };
__device__ __forceinline__ Pt unpack_xy(int2 v) { Pt p; p.lx = (int)(short)v.x; p.ly = v.x >> 16; p.rx = (int)(short)v.y; p.ry = v.y >> 16; return p; }

// Rank (0 = largest) of this thread's key among keys[0..n); keys are unique.  Caller syncs before (keys written) and after.
__device__ __forceinline__ int rank_of(const u64* keys, int n, u64 mine)
{
    int r = 0;
#pragma unroll 8
    for (int k = 0; k < n; ++k) r += keys[k] > mine;
    return r;
}

// sort the candidates by score (descending, ties by index): order[rank] = index.
template <class SM>
__device__ __forceinline__ void sort_scores(SM& sm, int num)
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
template <int NMAX, class SM, class Compat>
__device__ void greedy(SM& sm, int num, double thr, Compat compatible)
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
                li[u] = sm.li[idx[u]]; ri[u] = sm.ri[idx[u]];
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
template <class SM>
__device__ int compact(SM& sm)
{
    const int n = sm.nsel;
    const int t = threadIdx.x;
    float sim = 0, lo = 0, ro = 0; short li = 0, ri = 0; int2 xy = make_int2(0, 0);
    if (t < n) {
        const int s = sm.sel[t];
        sim = sm.sim[s]; li = sm.li[s]; ri = sm.ri[s]; xy = sm.xy[s]; lo = sm.lo[s]; ro = sm.ro[s];
    }
    __syncthreads();
    if (t < n) { sm.sim[t] = sim; sm.li[t] = li; sm.ri[t] = ri; sm.xy[t] = xy; sm.lo[t] = lo; sm.ro[t] = ro; }
    __syncthreads();
    return n;
}

// sum of cc[0..num) in ascending order; cc is zero-padded to a multiple of 4 (x + 0.0f == x)
template <class SM>
__device__ __forceinline__ float seq_sum(const SM& sm, int num)
{
    float sum = 0.0f;
    const float4* c4 = reinterpret_cast<const float4*>(sm.cc);
#pragma unroll 4
    for (int k = 0; k < (num + 3) / 4; ++k) { const float4 v = c4[k]; sum += v.x; sum += v.y; sum += v.z; sum += v.w; }
    return sum;
}

// |dist_latent - dist_rolled| of a correspondence pair; false when the pair is out of the look-up table's range (H = 0).
// LOOKUP: matcher.cpp:1246-1264 (block coordinates, table_dist).  else: :1372-1385 (pixels, sqrtf).
template <bool LOOKUP>
__device__ __forceinline__ bool pair_dist(const Pt& a, const Pt& o, const float* s_table, float& dist)
{
    float d1, d2; bool ok = true;
    if (LOOKUP) {
        const int dx1 = abs(a.lx - o.lx), dx2 = abs(a.rx - o.rx), dy1 = abs(a.ly - o.ly), dy2 = abs(a.ry - o.ry);
        ok = !((dx1 >= kDistN) | (dx2 >= kDistN) | (dy1 >= kDistN) | (dy2 >= kDistN));              // :1257
        d1 = s_table[ok ? dx1 * kDistN + dy1 : 0];
        d2 = s_table[ok ? dx2 * kDistN + dy2 : 0];
    } else {
        const float dx1 = (float)(a.lx - o.lx), dx2 = (float)(a.rx - o.rx), dy1 = (float)(a.ly - o.ly), dy2 = (float)(a.ry - o.ry);
        const float p = dx1 * dx1, q = dy1 * dy1, r = dx2 * dx2, s = dy2 * dy2;
        d1 = __fsqrt_rn(p + q);                                                                      // correctly rounded, as sqrtf
        d2 = __fsqrt_rn(r + s);
    }
    dist = fabsf(d1 - d2);
    return ok;
}
// H = clamp((30 - dist)/(25.0), 0, 1) for dist <= 30 (matcher.cpp:1268-1272 / :1389-1393): float numerator, double divide,
// float store.  (float)((double)x/25.0) == x/25.0f (double rounding through 53 bits is innocuous for a quotient of two
// 24-bit values), and for every float x in [0, 30] the fma sequence below equals x/25.0f — checked exhaustively over all
// 1,106,247,681 such floats by tools/verify_div25.c.
__device__ __forceinline__ float h_value(float dist)
{
    const float x = 30.0f - dist;
    const float q0 = x * 0.04f;
    const float r = fmaf(-q0, 25.0f, x);
    float h = fmaf(r, 0.04f, q0);
    if (h > 1.0f) h = 1.0f; else if (h < 0.0f) h = 0.0f;
    return h;
}

// S8a (LOOKUP = false, 5 iterations) / S8b (LOOKUP = true, 3 iterations).
template <int NMAX, int CACHE, bool LOOKUP, int ITERS>
__device__ int dist_filter(GraphSmem<NMAX, CACHE>& sm, int num, const float* s_table)
{
    typedef GraphSmem<NMAX, CACHE> SM;
    const int t = threadIdx.x;
    PHASE_INIT();
    for (int i = t; i < num * SM::W; i += kTailThreads) sm.hb[i / SM::W][i % SM::W] = 0u;
    if (t < SM::N4) { sm.b[t] = t < num ? sm.sim[t] : 0.0f; sm.cc[t] = 0.0f; }
    Pt me = {0, 0, 0, 0};
    if (t < num) me = unpack_xy(sm.xy[t]);
    __syncthreads();
    // non-zero pattern of the compatibility matrix (matcher.cpp:1237-1275 / :1363-1397): thread t visits the pairs
    // (t, t+d mod num), d = 1..num/2, so every unordered pair is evaluated once.  H != 0  <=>  in range and dist < 30.
    if (t < num) {
        const int half = num >> 1;
#pragma unroll 2
        for (int d = 1; d <= half; ++d) {
            if (d == half && !(num & 1) && t >= half) break;        // even num: the antipodal pairs are owned by the lower half
            int k = t + d; if (k >= num) k -= num;
            float dist;
            const bool ok = pair_dist<LOOKUP>(me, unpack_xy(sm.xy[k]), s_table, dist);
            if (ok && dist < 30.0f) {
                atomicOr(&sm.hb[t][k >> 5], 1u << (k & 31));
                atomicOr(&sm.hb[k][t >> 5], 1u << (t & 31));
            }
        }
    }
    __syncthreads();
    PHASE(8);
    uint32_t row[SM::W];
#pragma unroll
    for (int w = 0; w < SM::W; ++w) row[w] = t < num ? sm.hb[t][w] : 0u;
    // power iteration, :1284-1289 / :1406-1411 (canonical order: k ascending, unfused; see oracle)
    for (int it = 0; it < ITERS; ++it) {
        if (t < num) {
            float acc = 0.0f;
            int n = 0;
#pragma unroll
            for (int w = 0; w < SM::W; ++w) {
                uint32_t bits = row[w];
                while (bits) {
                    const int k = w * 32 + __ffs(bits) - 1;
                    bits &= bits - 1;
                    float h;
                    if (it == 0 || n >= CACHE) {
                        float dist;
                        pair_dist<LOOKUP>(me, unpack_xy(sm.xy[k]), s_table, dist);
                        h = h_value(dist);
                        if (n < CACHE) sm.stash[n * NMAX + t] = h;
                    } else h = sm.stash[n * NMAX + t];
                    const float p = h * sm.b[k];
                    acc += p;
                    ++n;
                }
            }
            sm.cc[t] = acc;
        }
        __syncthreads();
        const float sum = seq_sum(sm, num);
        const float scale = (float)(1.0 / ((double)sum + 0.00001));
        if (t < num) sm.b[t] = sm.cc[t] * scale;          // nobody reads b between the barrier above and the one below
        __syncthreads();
    }
    PHASE(9);
    sort_scores(sm, num);
    PHASE(10);
    greedy<NMAX>(sm, num, 0.0001, [&sm, s_table](int a, int o) {
        if (!((sm.hb[a][o >> 5] >> (o & 31)) & 1u)) return false;      // H == 0 < 1e-5
        float dist;
        pair_dist<LOOKUP>(unpack_xy(sm.xy[a]), unpack_xy(sm.xy[o]), s_table, dist);
        return !((double)h_value(dist) < 0.00001);
    });
    PHASE(11);
    const int nc = compact(sm);
    PHASE(12);
    return nc;
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

// the three angle tests of matcher.cpp:1495-1549 for the ordered pair (1 = lower index, 2 = higher index)
__device__ __forceinline__ bool angle_compatible(const Pt& p1, float lo1, float ro1, const Pt& p2, float lo2, float ro2)
{
    float angle_1 = adjust_angle(lo1 - lo2);
    float angle_2 = adjust_angle(ro1 - ro2);
    float angle_diff = fold_pi(fabsf(angle_1 - angle_2));
    if ((double)angle_diff > AFIS_PI / 4.) return false;
    const float dx_1 = (float)(p1.lx - p2.lx), dy_1 = (float)(p1.ly - p2.ly);
    const float line_angle_1 = -atan2_f32(dy_1, dx_1);
    angle_1 = adjust_angle(lo1 - line_angle_1);
    const float dx_2 = (float)(p1.rx - p2.rx), dy_2 = (float)(p1.ry - p2.ry);
    const float line_angle_2 = -atan2_f32(dy_2, dx_2);
    angle_2 = adjust_angle(ro1 - line_angle_2);
    angle_diff = fold_pi(fabsf(angle_1 - angle_2));
    if ((double)angle_diff > AFIS_PI / 6.) return false;
    angle_1 = adjust_angle(lo2 - line_angle_1);
    angle_2 = adjust_angle(ro2 - line_angle_2);
    angle_diff = fold_pi(fabsf(angle_1 - angle_2));
    if ((double)angle_diff > AFIS_PI / 6.) return false;
    return true;
}

// S9, matcher.cpp:1471-1636
template <int NMAX, int CACHE>
__device__ int angle_filter(GraphSmem<NMAX, CACHE>& sm, int num)
{
    typedef GraphSmem<NMAX, CACHE> SM;
    const int t = threadIdx.x;
    PHASE_INIT();
    for (int i = t; i < num * SM::W; i += kTailThreads) sm.hb[i / SM::W][i % SM::W] = 0u;
    if (t < SM::N4) { sm.b[t] = t < num ? (float)(1.0 / num) : 0.0f; sm.cc[t] = 0.0f; }   // :1558
    __syncthreads();
    if (t < num) {
        const Pt me = unpack_xy(sm.xy[t]);
        const float mlo = sm.lo[t], mro = sm.ro[t];
        const int half = num >> 1;
        for (int d = 1; d <= half; ++d) {
            if (d == half && !(num & 1) && t >= half) break;
            int k = t + d; if (k >= num) k -= num;
            const Pt o = unpack_xy(sm.xy[k]);
            const float olo = sm.lo[k], oro = sm.ro[k];
            const bool c = t < k ? angle_compatible(me, mlo, mro, o, olo, oro) : angle_compatible(o, olo, oro, me, mlo, mro);
            if (c) {
                atomicOr(&sm.hb[t][k >> 5], 1u << (k & 31));
                atomicOr(&sm.hb[k][t >> 5], 1u << (t & 31));
            }
        }
    }
    __syncthreads();
    PHASE(16);
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
    PHASE(17);
    sort_scores(sm, num);
    greedy<NMAX>(sm, num, 0.001, [&sm](int a, int o) { return (sm.hb[a][o >> 5] >> (o & 31)) & 1u; });
    PHASE(18);
    const int nc = compact(sm);
    PHASE(19);
    return nc;
}

template <class SM>
__device__ __forceinline__ float sum_sims(const SM& sm, int n)   // :508-514 / :775-781
{
    float score = 0.0f;
    for (int i = 0; i < n; ++i) score += sm.sim[i];
    return score;
}

// both graph stages; a list of fewer than 2 correspondences cannot survive S9 (a single node ends with S = 0)
template <int NMAX, int CACHE, bool LOOKUP, int ITERS>
__device__ __forceinline__ float graph_score(GraphSmem<NMAX, CACHE>& sm, int num, const float* s_table)
{
    num = dist_filter<NMAX, CACHE, LOOKUP, ITERS>(sm, num, s_table);
    if (num < 2) return 0.0f;
    num = angle_filter<NMAX, CACHE>(sm, num);
    return sum_sims(sm, num);
}

// =====================================================================================================================
// texture tail
// =====================================================================================================================
constexpr int kTexCache = 24;
struct TexSmem {
    GraphSmem<kTopTex, kTexCache> g;
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
            if (t < num) { sm.g.sim[r] = sm.tval[t]; sm.g.li[r] = sm.te[t]; sm.g.ri[r] = sm.targ[t]; }
        } else {                                                         // :748-749 rows stay in index order
            num = n_lt;
            if (t < num) { sm.g.sim[t] = rm_val[o + t]; sm.g.li[t] = (short)t; sm.g.ri[t] = (short)rm_arg[o + t]; }
        }
        __syncthreads();
        if (t < num) {
            const short2 lp = q.lt_xy[l0 + sm.g.li[t]], rp = g.tex_xy[r0 + sm.g.ri[t]];
            sm.g.xy[t] = pack_xy(lp.x, lp.y, rp.x, rp.y);
            sm.g.lo[t] = q.lt_ori[l0 + sm.g.li[t]]; sm.g.ro[t] = g.tex_ori[r0 + sm.g.ri[t]];
        }
        __syncthreads();
        const float score = graph_score<kTopTex, kTexCache, true, 3>(sm.g, num, sm.table);   // :759, :767
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
constexpr int kMinuCache = 16;
constexpr int kGemmRows = 64;       // latent rows per GEMM tile
constexpr int kGemmCols = 32;       // rolled columns per GEMM tile
constexpr int kGemmLd = 100;        // padded row stride (floats): 16-byte aligned rows, conflict-free b128 column walks
constexpr int kKeyRegs = 16;        // per-thread keys held in registers when nL*nR <= 256*16
constexpr int kFastL = 64, kFastR = 128;          // pair shapes whose whole similarity matrix stays in LDS
constexpr int kFastN = kFastL * kFastR;

struct MinuSmem {
    union {
        struct { float A[kGemmRows * kGemmLd]; float B[kGemmCols * kGemmLd]; } t;    // 38.4 KB, GEMM phase
        GraphSmem<kTopMinu, kMinuCache> g;                                            // graph phase
    } u;
    float simi[kFastN];             // 32 KB
    float rowsum[kFastL];
    float colsum[kFastR];
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

// Global scratch of one workgroup (pairs too large for the LDS fast path): simi[n] | keys[n] | rowsum[2000] | colsum[2000]
__global__ __launch_bounds__(kTailThreads) void k_minutiae(QueryDev q, GalleryDev g, float* __restrict__ scratch, size_t scratch_per_wg,
                                                           float* __restrict__ parts)
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
        const int qi = qs / 3, s = qs - qi * 3;
        const int l0 = q.lm_off[qs], nL = q.lm_off[qs + 1] - l0;
        const int r0 = g.minu_off[gi], nR = g.minu_off[gi + 1] - r0;
        float* out = parts + ((size_t)qi * g.G + gi) * 4 + s;
        if (nL <= 0 || nR <= 0) { if (tid == 0) *out = 0.0f; continue; }     // matcher.cpp:400-404
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
            for (int e = tid; e < kGemmRows * (kDes / 4); e += kTailThreads) {
                const int r = e / (kDes / 4), k4 = e - r * (kDes / 4);
                float4 a = make_float4(0, 0, 0, 0);
                if (it + r < nL) a = *reinterpret_cast<const float4*>(q.lm_des + (size_t)(l0 + it + r) * kDes + k4 * 4);
                *reinterpret_cast<float4*>(&sm.u.t.A[r * kGemmLd + k4 * 4]) = a;
            }
            for (int jt = 0; jt < nR; jt += kGemmCols) {
                if (jt) __syncthreads();
                for (int e = tid; e < kGemmCols * (kDes / 4); e += kTailThreads) {
                    const int r = e / (kDes / 4), k4 = e - r * (kDes / 4);
                    float4 b = make_float4(0, 0, 0, 0);
                    if (jt + r < nR) b = *reinterpret_cast<const float4*>(g.minu_des + (size_t)(r0 + jt + r) * kDes + k4 * 4);
                    *reinterpret_cast<float4*>(&sm.u.t.B[r * kGemmLd + k4 * 4]) = b;
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
                    for (int r = 0; r < 2; ++r) a[r] = *reinterpret_cast<const float4*>(&sm.u.t.A[(ty + 32 * r) * kGemmLd + k4 * 4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(&sm.u.t.B[(tx + 8 * c) * kGemmLd + k4 * 4]);
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
        for (int j = tid; j < nR; j += kTailThreads) {
            float sacc = 0.f;
#pragma unroll 8
            for (int i = 0; i < nL; ++i) sacc += simi[(size_t)i * nR + j];
            colsum[j] = sacc;
        }
        for (int i = tid; i < nL; i += kTailThreads) {
            float sacc = 0.f;
#pragma unroll 8
            for (int j = 0; j < nR; ++j) sacc += simi[(size_t)i * nR + j];
            rowsum[i] = sacc;
        }
        __syncthreads();
        PHASE(1);
        // ---- S3: top-120 by normalised similarity (:461-488) ----
        GraphSmem<kTopMinu, kMinuCache>& gs = sm.u.g;
        const int topN = n < kTopMinu ? n : kTopMinu;
        auto norm_key = [&](int e) {
            const int i = e / nR, j = e - i * nR;
            const float sv = simi[e];
            float f = rowsum[i] + colsum[j];
            f = f - sv;
            return ord_f32((float)((double)sv / ((double)f + 0.000001)));                        // :467
        };
        if (n <= kTailThreads * kKeyRegs) {                              // keys in registers
            uint32_t rk[kKeyRegs];
#pragma unroll
            for (int u = 0; u < kKeyRegs; ++u) { const int e = tid + u * kTailThreads; rk[u] = e < n ? norm_key(e) : 0u; }
            PHASE(2);
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
        } else {                                                         // large pairs: keys in the global scratch
            for (int e = tid; e < n; e += kTailThreads) gkeys[e] = norm_key(e);
            __syncthreads();
            select_topk(n, topN, [gkeys](int e) { return gkeys[e]; }, gs.keys, sm.te, gs.slots, parity, &gs.counter);
        }
        PHASE(3);
        if (tid < topN) {
            const int r = rank_of(gs.keys, topN, gs.keys[tid]);
            const int e = sm.te[tid];
            const int i1 = e / nR, i2 = e - i1 * nR;
            gs.sim[r] = simi[e]; gs.li[r] = (short)i1; gs.ri[r] = (short)i2;
            const short2 lp = q.lm_xy[l0 + i1], rp = g.minu_xy[r0 + i2];
            gs.xy[r] = pack_xy(lp.x, lp.y, rp.x, rp.y);
            gs.lo[r] = q.lm_ori[l0 + i1]; gs.ro[r] = g.minu_ori[r0 + i2];
        }
        __syncthreads();
        PHASE(4);
        const float score = graph_score<kTopMinu, kMinuCache, false, 5>(gs, topN, nullptr);   // :492, :495
        if (tid == 0) *out = score;
        __syncthreads();
        PHASE(5);
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
