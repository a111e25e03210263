// graph.hip — the geometric-consistency stages of a (latent, rolled) pair, one WAVE per correspondence list:
//   S7  top-200 texture rows (matching/matcher.cpp:736-749)
//   S8a LSS_R_Fast2_Dist_eigen  (matcher.cpp:1350-1469; pixels, sqrtf, 5 power iterations)  — minutiae lists, <= 120 entries
//   S8b LSS_R_Fast2_Dist_lookup (matcher.cpp:1225-1348; blocks, table_dist, 3 iterations)   — texture lists,  <= 200 entries
//   S9  LSS_R_Fast2 + adjust_angle (matcher.cpp:1471-1647)
//   and the final sum of the surviving similarities (matcher.cpp:508-514, :775-781).
//
// Why one wave per list: the work is a chain of short, partly serial phases (sorts, a greedy clique selection, sequential
// float sums).  A 64-lane workgroup needs no barriers, never has idle waves, and at ~20 KB of LDS eight of them fit on a CU,
// which is what hides the LDS / transcendental latency of each chain (a wave issues at most one VALU instruction per ~8 cycles on
// this chip; a SIMD needs >= 4 resident waves to stay busy).  Lane l owns rows l, l+64, ... of the list.
//
// Every float reduction keeps the reference's sequential order (index ascending, product and sum rounded separately; the
// file is compiled with -ffp-contract=off); where the reference's order is Eigen's (unpinned) the canonical order of
// oracle/afis_oracle.cpp is used.  Equal sort keys are ordered by ascending index.
//
// How the reference's serial steps are mapped without changing their results:
//   * std::sort of <= 200 keys        -> rank by counting (each key counts the keys larger than itself);
//   * "top 200 of n <= 1000"          -> bit-by-bit search for the 200th largest key with ballot/popcount (all keys in registers);
//   * greedy clique selection          -> rounds: the first still-alive candidate in rank order is accepted and every later
//                                        candidate that conflicts with it is killed in parallel; identical to the sequential
//                                        scan, but the trip count is the number of ACCEPTED candidates;
//   * H (200x200 fp32 = 160 KB)        -> never stored: LDS keeps the bitmask of its non-zero entries, values are recomputed on
//                                        demand and cached per row; a zero entry would only add +0.0f, so skipping it is exact.
#include "afis_device.h"
#include <type_traits>
#include "atan2f_libm.h"
#include "graph_arith.h"
#include "stdsort_order.h"

namespace afis {

#define AFIS_PI 3.1415926   /* matching/include.h:22 — a double literal; comparisons against it are in double */
typedef unsigned long long u64;

// optional per-phase cycle accounting (make PHASE_TIMING=1): slots 0..7 minutiae lists, 8..15 texture lists.  The stopwatch values are
// accumulated in LDS (WaveSmem::ph) and flushed with one global atomic per slot at the end of the kernel: a global atomic inside
// the loop stalls the global loads that follow it and distorts the phase it ends (round 1's timer did exactly that to the angle stage).
#ifdef AFIS_PHASE_TIMING
__device__ u64 g_graph_phase[16];
#define GPH_INIT() u64 gph_t0 = __builtin_readcyclecounter()
#define GPH(i) do { if (threadIdx.x == 0) { const u64 t1_ = __builtin_readcyclecounter(); sm.ph[(i)] += t1_ - gph_t0; gph_t0 = t1_; } } while (0)
#define GPH_T0() const u64 gph_k0 = __builtin_readcyclecounter()
#define GPH_K(i) do { if (threadIdx.x == 0) sm.ph[(i)] += __builtin_readcyclecounter() - gph_k0; } while (0)
#define GPH_ZERO() do { if (threadIdx.x < 16) sm.ph[threadIdx.x] = 0; WSYNC(); } while (0)
#define GPH_FLUSH() do { WSYNC(); if (threadIdx.x < 16 && sm.ph[threadIdx.x]) atomicAdd(&g_graph_phase[threadIdx.x], sm.ph[threadIdx.x]); } while (0)
#else
#define GPH_INIT() do {} while (0)
#define GPH(i) do {} while (0)
#define GPH_T0() do {} while (0)
#define GPH_K(i) do {} while (0)
#define GPH_ZERO() do {} while (0)
#define GPH_FLUSH() do {} while (0)
#endif

__device__ __forceinline__ uint32_t g_ord_f32(float v)
{
    v = v + 0.0f;                                   // -0 -> +0 so that equal floats get equal keys
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ u64 g_make_key(float v, int idx) { return ((u64)g_ord_f32(v) << 32) | (uint32_t)(~(uint32_t)idx); }
__device__ __forceinline__ int g_wave_popc(bool p) { return __popcll(__ballot(p)); }
__device__ __forceinline__ int g_lane_prefix(u64 mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0)); }
// single-wave workgroup: orders this wave's LDS traffic (s_barrier is free for one wave)
#define WSYNC() __syncthreads()

// maximum of x over the wave (wave-uniform result): DPP within the rows of 16 lanes, row_bcast across them, lane 63 holds the result
__device__ __forceinline__ int g_wave_max(int x)
{
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false));          // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false));          // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false));         // row_ror:4
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false));         // row_ror:8
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));         // row_bcast:15
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false));         // row_bcast:31
    return __builtin_amdgcn_readlane(x, 63);
}

// sum of x over the wave (wave-uniform result), same DPP ladder as g_wave_max
__device__ __forceinline__ int g_wave_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);                 // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);                 // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false);                // row_ror:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);                // row_ror:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);                // row_bcast:15 (rows 1, 3)
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);                // row_bcast:31 (rows 2, 3)
    return __builtin_amdgcn_readlane(x, 63);
}

struct Pt { int lx, ly, rx, ry; };
__device__ __forceinline__ int2 pack_xy(int lx, int ly, int rx, int ry) { return make_int2((lx & 0xffff) | (ly << 16), (rx & 0xffff) | (ry << 16)); }
__device__ __forceinline__ Pt unpack_xy(int2 v) { Pt p; p.lx = (int)(short)v.x; p.ly = v.x >> 16; p.rx = (int)(short)v.y; p.ry = v.y >> 16; return p; }

// The list's similarities: kept in LDS (minutiae lists: they exist nowhere else), or re-read from where they came from (texture lists:
// sim[t] is the row maximum of latent row li[t], still in L2 — 800 B less LDS is what lets a twelfth texture list share a CU).
template <int N, bool OWN> struct SimStore { float simv[N]; };
template <int N> struct SimStore<N, false> {};

#ifndef AFIS_MINU_ALIAS
#define AFIS_MINU_ALIAS 1
#endif
template <int NMAX_, int CACHE_, bool OWN_SIM_>
struct __attribute__((aligned(16))) WaveSmem : SimStore<NMAX_, OWN_SIM_> {
    static constexpr int NMAX = NMAX_, CACHE = CACHE_;
    static constexpr bool OWN_SIM = OWN_SIM_;
    static constexpr bool ALIAS_ORI = NMAX_ > 128 || AFIS_MINU_ALIAS;   // LDS is what limits the lists per CU: the angle stage's orientations live in b[] / cc[] while its H is built
    static constexpr int W = (NMAX + 31) / 32;
    static constexpr int N4 = (NMAX + 3) / 4 * 4;
    static constexpr int U = (NMAX + 63) / 64;
    float b[N4];                           // 16-byte aligned: read as float4 broadcasts
    union {                                // cc lives during the power iterations, order / sel from the sort that follows them
        float cc[N4];
        struct { short order[NMAX]; short sel[NMAX]; } os;     // rank -> candidate index; accepted candidates
    } y;
    short li[NMAX], ri[NMAX];
    int2 xy[NMAX];                         // .x = latent point, .y = rolled point: (x, y) as two fp16 on the packed paths (graph_arith.h), x | y << 16 as 16-bit integers otherwise and after S8
    uint32_t hb[NMAX][W];                  // bit rows: non-zero pattern of H (distance stage), then the boolean H of the angle stage
    // LDS is what limits how many lists a CU works on at once, so buffers with disjoint lifetimes share storage:
    union {
        float stash[CACHE * NMAX];                                                  // power iterations: CACHE * 4 bytes per row for the rows' first neighbour indices (bytes, [n][t]) and, optionally, values
        struct { u64 keys[N4]; float lo[ALIAS_ORI ? 1 : NMAX], ro[ALIAS_ORI ? 1 : NMAX]; } s;   // sorts (after the iterations); orientations (angle stage, unless they borrow b / cc)
        struct { uint32_t keys[N4]; short te[NMAX], targ[NMAX]; } pick;          // texture rows picked by S7, before they are ranked (32-bit keys)
    } x;
#ifdef AFIS_PHASE_TIMING
    u64 ph[16];
#endif
};
// Texture lists (OWN_SIM false) carry, in the spare bits of li / ri (latent row and rolled point are both < 1024: the 1000-row clamp, matcher.cpp:544-547), the SLOT of the row in the
// pair's array of row maxima: li = row | (slot & 31) << 10, ri = point | (slot >> 5) << 10.  With adc_variant 9 that array is the compact list the recomputation kernel wrote (the rows
// that can be among the 200, side by side in row order), otherwise the dense one (slot = row).  Minutiae lists hold plain indices.
template <class SM> __device__ __forceinline__ int l_row(int v) { if constexpr (SM::OWN_SIM) return v; else return v & 1023; }
template <class SM> __device__ __forceinline__ int r_pt(int v) { if constexpr (SM::OWN_SIM) return v; else return v & 1023; }
__device__ __forceinline__ int tex_slot_of(int li, int ri) { return ((li >> 10) & 31) | (((ri >> 10) & 31) << 5); }
__device__ __forceinline__ short tex_pack_l(int row, int slot) { return (short)(row | ((slot & 31) << 10)); }
__device__ __forceinline__ short tex_pack_r(int pt, int slot) { return (short)(pt | ((slot >> 5) << 10)); }
// similarity of list entry t; ext: the texture list's row maxima (the pair's compact or dense array), indexed by the entry's slot
template <class SM>
__device__ __forceinline__ float list_sim(const SM& sm, const float* __restrict__ ext, int t)
{
    if constexpr (SM::OWN_SIM) return sm.simv[t];
    else return ext[tex_slot_of(sm.li[t], sm.ri[t])];
}

// The wave's next task: a ticket from a global counter, one draw per list (a list costs 50-2000 us).  Every lane takes part in the
// atomic — lane 0 adds 1, the others 0, and the value lane 0 gets back is the ticket — so that there is NO divergent branch here.
// The usual "t = 0; if (lane == 0) t = atomicAdd(..); t = readfirstlane(t)" is not safe inside this loop: the compiler threads the
// lanes != 0 through a copy of the loop head in which t is the constant 0 (readfirstlane of a constant folds), and a task whose
// list is empty (`continue`) then spins forever; handing the ticket over through LDS behind wave barriers livelocks the same way
// (both were observed on the device).  So the single-lane atomic is one opaque asm block — exec narrowed to lane 0 around the
// instruction — which is also cheaper than what the backend makes of a 64-lane atomicAdd(ctr, lane == 0) (a 64-step scalar loop).
__device__ __forceinline__ int next_task(int32_t* ctr)
{
    int t;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(t), "=&s"(saved) : "v"(0), "v"(1), "s"(ctr) : "memory");
    return __builtin_amdgcn_readfirstlane(t);                            // all lanes are active again: lane 0 holds the ticket
}

// sort the candidates by score b (descending, ties by index): order[rank] = index.
// The ordered 32-bit keys are dealt into 64 bins of equal width between the smallest and the largest (counting sort through LDS counters, highest bin
// first); a key counts the larger keys of ITS bin only and adds the bins above it.  Equal scores make ranks collide, which the sum of the ranks shows (a
// permutation of 0 .. num-1 sums to num (num-1) / 2): the (key, ~index) composites are ranked against every other one only then.
// ref_tie (option ref_tie_order 2): where scores that can still be SELECTED (>= thr: the greedy walk stops at the first one below) tie, the order is the one libstdc++'s std::sort
// leaves (matcher.cpp:1301 / :1423 / :1590 sort the indices with a non-strict comparator) instead of ascending index: lane 0 runs the restatement of stdsort_order.h on the whole
// array (rare lists only: equal scores below thr — isolated candidates, all zero — do not count, and up to 16 entries std::sort is an insertion sort, which IS the ascending order).
template <class SM, int ref_tie>
__device__ __forceinline__ void sort_scores(SM& sm, int num, double thr)
{
    const int lane = threadIdx.x;
    constexpr int U = SM::U;
    uint32_t* const s_key = reinterpret_cast<uint32_t*>(sm.x.s.keys);         // [NMAX] keys in index order, then [128] bin fill pointers / starts
    uint32_t* const s_cnt = s_key + SM::N4;
    uint32_t* const s_gkey = reinterpret_cast<uint32_t*>(sm.y.cc);            // [NMAX] keys grouped by bin (cc is dead; order[] / sel[], which alias it, are written after the ranking)
    static_assert(sizeof(sm.x) >= (size_t)(SM::N4 + 128) * 4, "sort scratch exceeds the union");
    uint32_t m32[U]; int r[U];
    if (num <= 48) {                                                          // short lists (the angle stage's): the bins' bookkeeping costs more than it saves
        const u64 mine = lane < num ? g_make_key(sm.b[lane], lane) : 0ull;
        if (lane < num) sm.x.s.keys[lane] = mine;
        WSYNC();
        int rr = 0;
        for (int k = 0; k < num; ++k) rr += sm.x.s.keys[k] > mine;
        if (ref_tie && num > 16) {                                            // (uniform)
            int gg = 0;                                                       // the strictly larger SCORES: fewer than rr = an equal score with a lower index in front of this one
            for (int k = 0; k < num; ++k) gg += (uint32_t)(sm.x.s.keys[k] >> 32) > (uint32_t)(mine >> 32);
            const bool tied = lane < num && gg != rr && !((double)sm.b[lane] < thr);
            if (__ballot(tied) != 0ull) {
                uint32_t* const k32 = reinterpret_cast<uint32_t*>(sm.x.s.keys) + 96;          // behind the 48 composites: 48 keys, then the sort's stack
                static_assert(sizeof(sm.x.s.keys) >= (96 + 48 + 48) * 4, "no room for the keys and the stack of the std::sort restatement");
                WSYNC();
                if (lane < num) { k32[lane] = (uint32_t)(mine >> 32); sm.y.os.order[lane] = (short)lane; }
                WSYNC();
                if (lane == 0) stdsort_prefix<uint16_t>(reinterpret_cast<uint16_t*>(sm.y.os.order), num, num, k32, reinterpret_cast<int*>(k32 + 48));
                WSYNC();
                return;
            }
        }
        WSYNC();
        if (lane < num) sm.y.os.order[rr] = (short)lane;
        WSYNC();
        return;
    }
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = lane + 64 * u;
        m32[u] = t < num ? g_ord_f32(sm.b[t]) : 0u;
        if (t < num) { s_key[t] = m32[u]; kmin = min(kmin, m32[u]); kmax = max(kmax, m32[u]); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off)); }
    const uint32_t span = kmax - kmin;
    const int sh = max(0, 26 - __clz((int)(span | 1u)));                      // bin = (key - kmin) >> sh in [0, 63]
    int bin[U];
#pragma unroll
    for (int u = 0; u < U; ++u) bin[u] = (int)((m32[u] - kmin) >> sh) & 63;
    s_cnt[lane] = 0u;
    WSYNC();
#pragma unroll
    for (int u = 0; u < U; ++u) if (lane + 64 * u < num) atomicAdd(&s_cnt[bin[u]], 1u);
    WSYNC();
    {
        const int own = (int)s_cnt[lane];
        int suf = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_down(suf, off); if (lane + off < 64) suf += v; }
        WSYNC();
        s_cnt[lane] = (uint32_t)(suf - own); s_cnt[64 + lane] = (uint32_t)(suf - own);
    }
    WSYNC();
#pragma unroll
    for (int u = 0; u < U; ++u) if (lane + 64 * u < num) s_gkey[atomicAdd(&s_cnt[bin[u]], 1u)] = m32[u];
    WSYNC();
    int rsum = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        r[u] = 0;
        if (lane + 64 * u < num) {
            const int lo = (int)s_cnt[64 + bin[u]], hi = (int)s_cnt[bin[u]];
            int c = lo;
            for (int k = lo; k < hi; ++k) c += s_gkey[k] > m32[u];
            r[u] = c; rsum += c;
        }
    }
    if (g_wave_sum(rsum) != num * (num - 1) / 2) {                            // equal scores: rank the (key, ~index) composites
        u64 mine[U];
        int gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int t = lane + 64 * u; mine[u] = t < num ? ((u64)m32[u] << 32) | (uint32_t)(~(uint32_t)t) : 0ull; r[u] = 0; gg[u] = 0; }
        for (int k = 0; k < num; ++k) {
            const u64 kk = ((u64)s_key[k] << 32) | (uint32_t)(~(uint32_t)k);
#pragma unroll
            for (int u = 0; u < U; ++u) { r[u] += kk > mine[u]; gg[u] += s_key[k] > m32[u]; }
        }
        if (ref_tie) {                                                        // (uniform) a tie among scores that can be selected: the reference's order = std::sort's
            bool tied = false;
#pragma unroll
            for (int u = 0; u < U; ++u) { const int t = lane + 64 * u; tied |= t < num && gg[u] != r[u] && !((double)sm.b[t] < thr); }
            if (__ballot(tied) != 0ull) {
                WSYNC();                                                      // s_gkey (= cc, which order[] aliases) has been read by every lane
#pragma unroll
                for (int u = 0; u < U; ++u) { const int t = lane + 64 * u; if (t < num) sm.y.os.order[t] = (short)t; }
                WSYNC();
                if (lane == 0) stdsort_prefix<uint16_t>(reinterpret_cast<uint16_t*>(sm.y.os.order), num, num, s_key, reinterpret_cast<int*>(s_cnt));   // s_cnt: 128 words, dead; the stack takes 3 (2 log2 num + 1) <= 48
                WSYNC();
                return;
            }
        }
    }
    WSYNC();
#pragma unroll
    for (int u = 0; u < U; ++u) { const int t = lane + 64 * u; if (t < num) sm.y.os.order[r[u]] = (short)t; }
    WSYNC();
}

// Greedy selection, matcher.cpp:1304-1344 / :1425-1465 / :1593-1633: walk the candidates by descending S; stop at S < thr;
// skip a candidate whose latent or rolled point is already used or that is incompatible with ANY accepted one.
// The wave holds the candidates in rank order (lane l: ranks l, l+64, ...).  Each round accepts the first alive candidate and
// kills every later one that conflicts with it.  Accepted indices go to sm.y.os.sel[0..nsel) in acceptance (= rank) order.
template <class SM, class Compat>
__device__ int greedy(SM& sm, int num, double thr, Compat compatible)
{
    const int lane = threadIdx.x;
    constexpr int U = SM::U;
    int idx[U], li[U], ri[U];
    bool alive[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = lane + 64 * u;
        idx[u] = 0; li[u] = -1; ri[u] = -1; alive[u] = false;
        if (p < num) {
            idx[u] = sm.y.os.order[p];
            li[u] = l_row<SM>(sm.li[idx[u]]); ri[u] = r_pt<SM>(sm.ri[idx[u]]);
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
        if (lane == 0) sm.y.os.sel[nsel] = (short)cidx;
        ++nsel;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (alive[u]) {
                const int p = lane + 64 * u;
                if (p == first || li[u] == cli || ri[u] == cri || !compatible(cidx, idx[u])) alive[u] = false;
            }
        }
    }
    WSYNC();
    return nsel;
}

// keep only the accepted correspondences, in acceptance order
template <class SM>
__device__ void compact(SM& sm, int n)
{
    const int lane = threadIdx.x;
    float sim[SM::U]; short li[SM::U], ri[SM::U]; int2 xy[SM::U];
#pragma unroll
    for (int u = 0; u < SM::U; ++u) {
        const int t = lane + 64 * u;
        sim[u] = 0.0f;
        if (t < n) { const int s = sm.y.os.sel[t]; if constexpr (SM::OWN_SIM) sim[u] = sm.simv[s]; li[u] = sm.li[s]; ri[u] = sm.ri[s]; xy[u] = sm.xy[s]; }
    }
    WSYNC();
#pragma unroll
    for (int u = 0; u < SM::U; ++u) {
        const int t = lane + 64 * u;
        if (t < n) { if constexpr (SM::OWN_SIM) sm.simv[t] = sim[u]; sm.li[t] = li[u]; sm.ri[t] = ri[u]; sm.xy[t] = xy[u]; }
    }
    WSYNC();
}

// sum of cc[0..num) in ascending order; cc is zero-padded to a multiple of 4 (x + 0.0f == x)
template <class SM>
__device__ __forceinline__ float seq_sum(const SM& sm, int num)
{
    float sum = 0.0f;
    const float4* c4 = reinterpret_cast<const float4*>(sm.y.cc);
#pragma unroll 4
    for (int k = 0; k < (num + 3) / 4; ++k) { const float4 v = c4[k]; sum += v.x; sum += v.y; sum += v.z; sum += v.w; }
    return sum;
}

// |dist_latent - dist_rolled| of a correspondence pair; false when the pair is out of the look-up table's range (H = 0).
// LOOKUP: matcher.cpp:1246-1264 (block coordinates, table_dist).  else: :1372-1385 (pixels, sqrtf).
template <bool LOOKUP>
__device__ __forceinline__ bool pair_dist(const Pt& a, const Pt& o, const float* __restrict__ table, float& dist)
{
    float d1, d2; bool ok = true;
    if (LOOKUP) {
        const int dx1 = abs(a.lx - o.lx), dx2 = abs(a.rx - o.rx), dy1 = abs(a.ly - o.ly), dy2 = abs(a.ry - o.ry);
        ok = !((dx1 >= kDistN) | (dx2 >= kDistN) | (dy1 >= kDistN) | (dy2 >= kDistN));              // :1257
        // table_dist[dx*50+dy] = (float)sqrt((16 dx)^2 + (16 dy)^2) (matcher.cpp:45-56).  The argument is an exact integer
        // < 2^24, and a double sqrt rounded to float equals the correctly rounded float sqrt, so the table entry is
        // recomputed bit-exactly instead of being fetched (tests/test_host.py checks all 2500 entries).
        d1 = sqrt_rn_pos((float)(256 * (dx1 * dx1 + dy1 * dy1)));
        d2 = sqrt_rn_pos((float)(256 * (dx2 * dx2 + dy2 * dy2)));
        (void)table;
    } else {
        const float dx1 = (float)(a.lx - o.lx), dx2 = (float)(a.rx - o.rx), dy1 = (float)(a.ly - o.ly), dy2 = (float)(a.ry - o.ry);
        const float p = dx1 * dx1, q = dy1 * dy1, r = dx2 * dx2, s = dy2 * dy2;
        d1 = sqrt_rn_pos(p + q);                                                                     // correctly rounded, as sqrtf
        d2 = sqrt_rn_pos(r + s);
    }
    dist = fabsf(d1 - d2);
    return ok;
}
// "H != 0" for a pair, i.e. in range and |d1 - d2| < 30 with d1, d2 the correctly rounded distances — decided from the 1-ulp
// hardware square roots whenever the outcome cannot depend on their last bit: each d is within one ulp of its exact value, so
// |d1 - d2| is within 3 ulp(max(d1, d2)) <= max(d1, d2) * 2^-21 of the exact difference (incl. the subtraction's own rounding);
// only pairs closer than that to the threshold (about one in 10^5) take the exact evaluation.
template <bool LOOKUP>
__device__ __forceinline__ bool pair_compatible(const Pt& a, const Pt& o, const float* __restrict__ table)
{
    float s1, s2; bool ok = true;
    if (LOOKUP) {
        const int dx1 = abs(a.lx - o.lx), dx2 = abs(a.rx - o.rx), dy1 = abs(a.ly - o.ly), dy2 = abs(a.ry - o.ry);
        ok = !((dx1 >= kDistN) | (dx2 >= kDistN) | (dy1 >= kDistN) | (dy2 >= kDistN));              // :1257
        s1 = (float)(256 * (dx1 * dx1 + dy1 * dy1)); s2 = (float)(256 * (dx2 * dx2 + dy2 * dy2));
    } else {
        const float dx1 = (float)(a.lx - o.lx), dx2 = (float)(a.rx - o.rx), dy1 = (float)(a.ly - o.ly), dy2 = (float)(a.ry - o.ry);
        const float p = dx1 * dx1, q = dy1 * dy1, r = dx2 * dx2, s = dy2 * dy2;
        s1 = p + q; s2 = r + s;
    }
    const float d1 = __builtin_amdgcn_sqrtf(s1), d2 = __builtin_amdgcn_sqrtf(s2);
    const float dist = fabsf(d1 - d2);
    const float slack = fmaxf(d1, d2) * 4.76837158e-7f;                  // 2^-21
    if (fabsf(dist - 30.0f) > slack) return ok && dist < 30.0f;
    float exact;
    pair_dist<LOOKUP>(a, o, table, exact);
    return ok && exact < 30.0f;
}
// ---- texture lists whose block coordinates all lie in [0, 8191] (always, for real templates): the same two functions on the
// packed (x | y << 16) words as they sit in LDS.  One v_pk_sub_i16 gives (dx, dy), one v_dot2_i32_i16 gives n = dx^2 + dy^2
// (<= 4802 for an in-range pair), and table_dist[dx*50+dy] = RN(sqrt(256 n)) = 16 * RN(sqrt(n)) because scaling by a power of
// two commutes with every rounding involved: dist = |d1 - d2| = 16 * |RN(sqrt n1) - RN(sqrt n2)| exactly.
template <bool RANGE_TEST>
__device__ __forceinline__ bool tex_pair_dist(int2 a, int2 o, float& dist)
{
    float n1, n2;
    pair_n(a, o, n1, n2);
    dist = 16.0f * fabsf(sqrt_rn_int(n1) - sqrt_rn_int(n2));
    return RANGE_TEST ? tex_in_range(a, o) : true;
}
template <bool RANGE_TEST>
__device__ __forceinline__ bool tex_pair_compatible(int2 a, int2 o)
{
    float n1, n2;
    pair_n(a, o, n1, n2);
    const bool ok = RANGE_TEST ? tex_in_range(a, o) : true;              // out of range: n may exceed 2^24, the answer is masked
    return pair_compatible_n<true>(n1, n2) && ok;
}

// ---- minutiae lists whose pixel coordinates all lie in [0, 2047] (any image up to 2048 px): the reference's float arithmetic
// dx*dx + dy*dy (matcher.cpp:1372-1385) is exact there (each square < 2^22, the sum < 2^23), so it equals the integer
// v_dot2_i32_i16 of the packed differences.
__device__ __forceinline__ float minu_pair_dist(int2 a, int2 o)
{
    float n1, n2;
    pair_n(a, o, n1, n2);
    return fabsf(sqrt_rn_int(n1) - sqrt_rn_int(n2));
}
__device__ __forceinline__ bool minu_pair_compatible(int2 a, int2 o)
{
    float n1, n2;
    pair_n(a, o, n1, n2);
    return pair_compatible_n<false>(n1, n2);
}

// H = clamp((30 - dist)/(25.0), 0, 1) for dist <= 30 (matcher.cpp:1268-1272 / :1389-1393): float numerator, double divide,
// float store.  (float)((double)x/25.0) == x/25.0f (double rounding through 53 bits is innocuous for a quotient of two
// 24-bit values), and for every float x in [0, 30] the fma sequence below equals x/25.0f — checked exhaustively over all
// 1,106,247,681 such floats by tools/verify_div25.c.
__device__ __forceinline__ float h_of_x(float x)
{
    const float q0 = x * 0.04f;
    const float r = fmaf(-q0, 25.0f, x);
    const float h = fmaf(r, 0.04f, q0);
    return __builtin_amdgcn_fmed3f(h, 0.0f, 1.0f);
}
__device__ __forceinline__ float h_value(float dist)
{
    const float x = 30.0f - dist;
    const float q0 = x * 0.04f;
    const float r = fmaf(-q0, 25.0f, x);
    const float h = fmaf(r, 0.04f, q0);
    return __builtin_amdgcn_fmed3f(h, 0.0f, 1.0f);                       // h is never NaN here; clamp to [0, 1] in one instruction
}

// S8a (LOOKUP = false, 5 iterations) / S8b (LOOKUP = true, 3 iterations).  Returns the number of survivors (compacted in place).
// MODE 0: generic arithmetic; 1: packed 16-bit coordinates (texture: with the |d| < 50 test); 2: texture, every coordinate in [0, 49]
template <class SM, bool LOOKUP, int ITERS, int MODE, int REF_TIE>
__device__ int dist_filter(SM& sm, int num, const float* __restrict__ table, const float* __restrict__ ext)
{
    constexpr bool fast = MODE > 0;
    constexpr bool range_test = MODE == 1;
    auto compat_fast = [](int2 a, int2 o) -> bool { if (LOOKUP) return tex_pair_compatible<range_test>(a, o); return minu_pair_compatible(a, o); };
    auto dist_fast = [](int2 a, int2 o) -> float { float d; if (LOOKUP) { tex_pair_dist<range_test>(a, o, d); return d; } return minu_pair_dist(a, o); };
    constexpr int U = SM::U, W = SM::W, NMAX = SM::NMAX;
    [[maybe_unused]] constexpr int PH = LOOKUP ? 8 : 0;
    GPH_INIT();
    const int lane = threadIdx.x;
    for (int i = lane; i < num * W; i += 64) sm.hb[i / W][i % W] = 0u;
    int2 me[U];                                                           // own points, packed as in LDS
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = lane + 64 * u;
        if (t < SM::N4) { sm.b[t] = t < num ? list_sim(sm, ext, t) : 0.0f; sm.y.cc[t] = 0.0f; }
        me[u] = t < num ? sm.xy[t] : make_int2(0, 0);
    }
    WSYNC();
    // non-zero pattern of the compatibility matrix (matcher.cpp:1237-1275 / :1363-1397): row t visits the pairs
    // (t, t+d mod num), d = 1..num/2, so every unordered pair is evaluated once.  H != 0  <=>  in range and dist < 30.
    const int half = num >> 1;
    const bool even = !(num & 1);
    uint32_t* const hb0 = &sm.hb[0][0];
    auto pair = [&](int2 own, int t, int k, int2 other) {
        if (fast ? compat_fast(own, other) : pair_compatible<LOOKUP>(unpack_xy(own), unpack_xy(other), table)) {
            atomicOr(hb0 + (t * W + (k >> 5)), 1u << (k & 31));
            atomicOr(hb0 + (__umul24(k, W) + (t >> 5)), 1u << (t & 31));
        }
    };
    // The last block of 64 rows is usually far from full (a texture list has 200 rows: 8 in its fourth block).  When it holds <= 32
    // rows, its lanes are split into 64/RG groups that take different offsets d of the same RG rows, instead of idling.
    const int tail0 = num & ~63, R = num - tail0;
    const bool grouped = R > 0 && R <= 32;
    const int n_wide = grouped ? tail0 : num;                             // rows handled one per lane and block
    // t + d (mod num), kept incrementally as the BYTE offset of the partner point in xy[]: add, subtract, unsigned minimum (the difference
    // wraps to a huge number until the offset reaches the end, where it is 0) — two full-rate instructions and a minimum instead of
    // compare, select and the shift of the address (profiles/r03_valu_cost_table.json: shifts left, compares and selects cost twice an add)
    uint32_t kb[U];
    const uint32_t num8 = (uint32_t)num * 8u;
#pragma unroll
    for (int u = 0; u < U; ++u) { const int t = lane + 64 * u; kb[u] = t < num ? (uint32_t)t * 8u : 0u; }
#ifndef AFIS_PAIR_UNROLL
#define AFIS_PAIR_UNROLL 1
#endif
    constexpr int DU = fast ? AFIS_PAIR_UNROLL : 1;                       // offsets d handled per trip of the loop: DU x U independent chains per lane
    // UA = row blocks that hold one-per-lane rows (a texture list of 200 rows: 3 — its 8-row fourth block is taken by the lane groups below —
    // so a fourth chain would compute on nothing for the whole loop)
    auto pair_loop = [&](auto ua_tag) {
    constexpr int UA = decltype(ua_tag)::value;
    for (int d = 1; d <= half; d += DU) {
        int2 other[DU][UA];                                                // the partner points are fetched together: one LDS round trip per trip
        int kk[DU][UA];
#pragma unroll
        for (int s2 = 0; s2 < DU; ++s2) {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const uint32_t a = kb[u] + 8u;
                kb[u] = min(a, a - num8);
                other[s2][u] = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(sm.xy) + kb[u]);
                kk[s2][u] = (int)(kb[u] >> 3);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (fast) {
            // the predicates are evaluated without control flow between them (rows beyond the list compute on zeros), so their
            // instruction chains interleave; the rare band case of any of them is one uniform branch for all
            float n1[DU][UA], n2[DU][UA]; bool hit[DU][UA], near[DU][UA]; bool any_near = false;
#pragma unroll
            for (int s2 = 0; s2 < DU; ++s2) {
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    pair_n(me[u], other[s2][u], n1[s2][u], n2[s2][u]);
                    hit[s2][u] = pair_compatible_t<LOOKUP>(n1[s2][u], n2[s2][u], near[s2][u]);
                    hit[s2][u] = hit[s2][u] | near[s2][u];
                }
            }
            // A pair inside the guard band of the square-root-free test (graph_arith.h) is simply taken as a hit: every consumer of a set bit evaluates
            // the pair's exact value — the power iterations multiply by h (= 0 for a distance of 30 or a hair more: x + 0.0f == x), the greedy selection
            // tests h >= 1e-5 — so a spurious bit changes no result, only adds a zero-valued neighbour (1e-4 of the pairs at most).
            (void)any_near;
#pragma unroll
            for (int s2 = 0; s2 < DU; ++s2) {
                const bool last = d + s2 == half && even;                 // even num: the antipodal pairs belong to the lower half
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    const int t = lane + 64 * u, k = kk[s2][u];
                    bool ok = hit[s2][u] && t < n_wide && d + s2 <= half && !(last && t >= half);
                    if (LOOKUP && range_test) ok = ok && tex_in_range(me[u], other[s2][u]);
                    if constexpr (LOOKUP) {
                        // texture lists (three chains): the atomics are issued by every lane, with a zero operand where there is no hit — six full-width
                        // LDS atomics cost less than three divergent regions per trip (-2.8 %; minutiae lists, two chains: +1.2 %, they keep the branch)
                        const int ts = t < num ? t : 0;
                        atomicOr(hb0 + (ts * W + (k >> 5)), ok ? 1u << (k & 31) : 0u);
                        atomicOr(hb0 + (__umul24(k, W) + (ts >> 5)), ok ? 1u << (ts & 31) : 0u);
                    } else if (ok) {
                        atomicOr(hb0 + (t * W + (k >> 5)), 1u << (k & 31));
                        atomicOr(hb0 + (__umul24(k, W) + (t >> 5)), 1u << (t & 31));
                    }
                }
            }
        } else {
            const bool last = d == half && even;
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int t = lane + 64 * u;
                if (t < n_wide && !(last && t >= half)) pair(me[u], t, kk[0][u], other[0][u]);
            }
        }
    }
    };
    const int n_blk = (n_wide + 63) >> 6;
    if (n_blk >= U) pair_loop(std::integral_constant<int, U>{});
    else if (U > 1 && n_blk == U - 1) pair_loop(std::integral_constant<int, (U > 1 ? U - 1 : 1)>{});
    else if (n_blk > 0) pair_loop(std::integral_constant<int, U>{});
    if (grouped) {
        const int sh = R <= 8 ? 3 : R <= 16 ? 4 : 5, RG = 1 << sh, dstep = 64 >> sh;
        const int r = lane & (RG - 1), t = tail0 + r;
        const bool row = r < R;
        const int2 own = row ? sm.xy[t] : make_int2(0, 0);
        int d = 1 + (lane >> sh);
        int k = t + d; k = k >= num ? k - num : k;                        // d <= half < num: one wrap
        for (int d0 = 1; d0 <= half; d0 += dstep) {
            if (row && d <= half && !(d == half && even && t >= half)) pair(own, t, k, sm.xy[k]);
            d += dstep; k += dstep; k = k >= num ? k - num : k;           // exact while d <= half (k < num + dstep <= 2 num there)
        }
    }
    WSYNC();
    GPH(PH + 0);
    // Rows differ in their number of non-zeros, and a wave pays for the longest row of every pass.  So the rows are dealt to the
    // lanes in descending order of their non-zero count (16 buckets of 4, counting sort with ballots): the rows of one pass
    // then have nearly equal lengths.  Rows are independent, so the row -> lane assignment does not touch any result.
    int myrow[U], trip[U], rlen[U];
    const int Wn = (num + 31) / 32;
    {
        int cnt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = lane + 64 * u;
            int c = -1;
            if (t < num) {
                c = 0; for (int w = 0; w < Wn; ++w) c += __popc(sm.hb[t][w]);
                sm.y.os.sel[t] = (short)c;                                 // exact count (sel[] aliases cc[num/2 .. num): dead until the iterations)
                c = min(c >> 2, 15);
            }
            cnt[u] = c;
        }
        int base = 0;
        for (int bk = 15; bk >= 0; --bk) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = cnt[u] == bk;
                const u64 m = __ballot(in);
                if (in) sm.y.os.order[base + g_lane_prefix(m)] = (short)(lane + 64 * u);   // order[] aliases cc[0 .. num/2): dead until the iterations
                base += __popcll(m);
            }
        }
        WSYNC();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = lane + 64 * u;
            myrow[u] = p < num ? (int)sm.y.os.order[p] : -1;
            rlen[u] = myrow[u] >= 0 ? (int)sm.y.os.sel[myrow[u]] : 0;
            trip[u] = g_wave_max(rlen[u]);                                               // longest row of this pass: the pass's trip count
        }
        WSYNC();
        for (int t = num + lane; t < SM::N4; t += 64) sm.y.cc[t] = 0.0f;   // sel[] may have run over cc's zero padding (seq_sum reads it)
    }
    int2 mine[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mine[u] = myrow[u] >= 0 ? sm.xy[myrow[u]] : make_int2(0, 0);
    // power iteration, :1284-1289 / :1406-1411 (canonical order: k ascending, unfused; see oracle).
    // Iteration 0 walks each row's bit mask (every lane ITS row, a (word, remaining bits) cursor inside ONE loop whose trip count is the longest
    // row of the pass), computes every value and notes, per row, the first kIdxN neighbour indices as bytes and the first kValN values
    // (idx8[n][row], vst[n][row]: together they fill the stash space).  Iterations 1.. read the n-th neighbour from there — a byte load instead of
    // the bit walk with its divergent "next non-empty word" loop — take the value from the stash (n < kValN) or recompute it (20 instructions
    // in the fp16 form); only rows longer than kIdxN go on with the bit walk, from the cursor iteration 0 left at position kIdxN.
    // Texture lists (200 rows, 17 neighbours on average): 8 indices per row, minutiae lists (120 rows, 10 on average): 12, and no values — values + indices
    // were 2-4 % faster at equal occupancy, but LDS decides how many lists a SIMD holds, and two more texture lists per CU / two more minutiae lists per SIMD are worth more.
    // (Before the index lists: 4 and 10 values per row; the walk cost as much as a value.)
    constexpr bool kFlat = NMAX > 128;
#ifndef AFIS_MINU_VALN
#define AFIS_MINU_VALN 0
#endif
    constexpr int kValN = kFlat ? 0 : AFIS_MINU_VALN, kIdxN = 4 * SM::CACHE - 4 * kValN;   // 4 kValN + kIdxN bytes per row = the stash space (CACHE floats per row)
    static_assert(kValN * NMAX * 4 + kIdxN * NMAX <= (int)sizeof(sm.x.stash), "index lists + value stash exceed the stash space");
    float* const vst = sm.x.stash;
    unsigned char* const idx8 = reinterpret_cast<unsigned char*>(sm.x.stash + kValN * NMAX);
#ifndef AFIS_MINU_REGN
#define AFIS_MINU_REGN 0
#endif
    // minutiae lists only: the values of a row's first kRegN neighbours stay in REGISTERS from iteration 0 on (statically indexed: the first kRegN steps of every loop are unrolled)
    constexpr int kRegN = kFlat ? 0 : AFIS_MINU_REGN;
    static_assert(kRegN <= kIdxN, "cached values need their neighbour indices");
    float hreg[U][kRegN > 0 ? kRegN : 1];
    int cur_w[U]; uint32_t cur_bits[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { cur_w[u] = 0; cur_bits[u] = 0u; }
    auto value = [&](int2 own, int k) -> float {
        float dist;
        if (fast && LOOKUP) {
            // texture lists: dist = 16 |RN sqrt n1 - RN sqrt n2| (tex_pair_dist), and 30 - dist in ONE fma: the product by 16 is exact, so fl(30 - 16 d) is the same float either way
            // (one instruction less in the loop that is half of this kernel: 27 instead of 28 per neighbour)
            float n1, n2;
            pair_n(own, sm.xy[k], n1, n2);
            return h_of_x(fmaf(-16.0f, fabsf(sqrt_rn_int(n1) - sqrt_rn_int(n2)), 30.0f));
        }
        if (fast) dist = dist_fast(own, sm.xy[k]);
        else pair_dist<LOOKUP>(unpack_xy(own), unpack_xy(sm.xy[k]), table, dist);
        return h_value(dist);
    };
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = myrow[u];
            const uint32_t* hrow = sm.hb[t >= 0 ? t : 0];
            float acc = 0.0f;
            if (it == 0) {
                if (t >= 0) {                                                // lanes beyond the list sit the pass out: ONE divergent region, not one per step
                    int w = 0;
                    uint32_t bits = hrow[0];
#pragma unroll
                    for (int r = 0; r < kRegN; ++r) {                        // the first kRegN steps, unrolled: their values go to registers
                        if (r < trip[u]) {                                   // uniform
                            while (bits == 0u && w + 1 < Wn) { ++w; bits = hrow[w]; }
                            const bool have = bits != 0u;
                            const int k = (w * 32 + __ffs(bits) - 1) & 255;
                            bits &= bits - 1;
                            const float h = value(mine[u], k);
                            idx8[r * NMAX + t] = (unsigned char)k;
                            hreg[u][r] = h;
                            const float p = h * sm.b[k];
                            acc += have ? p : 0.0f;
                        }
                    }
                    for (int n = kRegN; n < trip[u]; ++n) {                  // uniform
                        if (n == kIdxN) { cur_w[u] = w; cur_bits[u] = bits; }    // uniform condition
                        while (bits == 0u && w + 1 < Wn) { ++w; bits = hrow[w]; }   // next non-empty word of this lane's row
                        const bool have = bits != 0u;                        // an ended row notes a stale index (never read: n >= its length) and adds +0.0f
                        const int k = (w * 32 + __ffs(bits) - 1) & 255;
                        bits &= bits - 1;
                        const float h = value(mine[u], k);
                        if (n < kIdxN) idx8[n * NMAX + t] = (unsigned char)k;
                        if (n < kValN) vst[n * NMAX + t] = h;
                        const float p = h * sm.b[k];
                        acc += have ? p : 0.0f;
                    }
                }
            } else {
                const int nV = min(trip[u], kValN), nA = min(trip[u], kIdxN);
                const int tt = t >= 0 ? t : 0;
                // no divergent region per step: a lane whose row has ended computes on a stale index byte (any byte addresses LDS of this list) and adds
                // +0.0f instead (acc >= +0: x + 0.0f == x) — the exec bookkeeping of a branch costs more than the masked lanes' work
                for (int n = 0; n < nV; ++n) {                               // uniform
                    const int k = idx8[n * NMAX + tt];
                    const float p = vst[n * NMAX + tt] * sm.b[k];
                    acc += n < rlen[u] ? p : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < kRegN; ++r) {
                    if (r < nA) {                                            // uniform
                        const int k = idx8[r * NMAX + tt];
                        const float p = hreg[u][r] * sm.b[k];
                        acc += r < rlen[u] ? p : 0.0f;
                    }
                }
                for (int n = max(nV, kRegN); n < nA; ++n) {
                    const int k = idx8[n * NMAX + tt];
                    const float p = value(mine[u], k) * sm.b[k];
                    acc += n < rlen[u] ? p : 0.0f;
                }
                if (trip[u] > kIdxN) {
                    int w = cur_w[u];
                    uint32_t bits = cur_bits[u];
                    for (int n = kIdxN; n < trip[u]; ++n) {
                        while (bits == 0u && w + 1 < Wn) { ++w; bits = t >= 0 ? hrow[w] : 0u; }
                        const bool have = bits != 0u;                        // as above: no divergent region around the value, an ended row adds +0.0f
                        const int k = (w * 32 + __ffs(bits) - 1) & 255;
                        bits &= bits - 1;
                        const float p = value(mine[u], k) * sm.b[k];
                        acc += have ? p : 0.0f;
                    }
                }
            }
            if (t >= 0) sm.y.cc[t] = acc;
        }
        WSYNC();
        const float sum = seq_sum(sm, num);
        const float scale = (float)(1.0 / ((double)sum + 0.00001));
#pragma unroll
        for (int u = 0; u < U; ++u) { const int t = lane + 64 * u; if (t < num) sm.b[t] = sm.y.cc[t] * scale; }
        WSYNC();
    }
    GPH(PH + 1);
    sort_scores<SM, REF_TIE>(sm, num, 0.0001);
    GPH(PH + 2);
    const int nsel = greedy(sm, num, 0.0001, [&sm, table, dist_fast](int a, int o) {
        if (!((sm.hb[a][o >> 5] >> (o & 31)) & 1u)) return false;      // H == 0 < 1e-5
        float dist;
        if (fast) dist = dist_fast(sm.xy[a], sm.xy[o]);
        else pair_dist<LOOKUP>(unpack_xy(sm.xy[a]), unpack_xy(sm.xy[o]), table, dist);
        return !((double)h_value(dist) < 0.00001);
    });
    compact(sm, nsel);
    GPH(PH + 3);
    return nsel;
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
// atan2f of the reference's CPU build = the C library's (matcher.cpp:1516, :1524).  glibc's routine is not correctly rounded (a
// correctly rounded atan2 differs from it by one ulp on 16 % of the integer coordinate differences), so the device evaluates
// the same fp32 operation sequence instead (atan2f_libm.h); tests compare the two exhaustively over [-2047, 2047]^2.
__device__ __forceinline__ float atan2_f32(float y, float x) { return afis_atan2f_libm(y, x); }

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

// S9, matcher.cpp:1471-1636.  Returns the number of survivors (compacted in place).  lori / rori: orientation arrays of the
// latent and rolled template (global memory), indexed by the correspondences' point indices; only the survivors of the distance
// stage need them.
template <class SM, int REF_TIE>
__device__ int angle_filter(SM& sm, int num, const float* __restrict__ lori, const float* __restrict__ rori)
{
    constexpr int W = SM::W;
    [[maybe_unused]] constexpr int PH = SM::NMAX > 128 ? 8 : 0;
    GPH_INIT();
    const int lane = threadIdx.x;
    // texture lists: the orientations are needed while the boolean H is built, b[] and cc[] only by the iterations after it: they share the storage
    if constexpr (SM::ALIAS_ORI) {
        for (int t = lane; t < num; t += 64) { sm.b[t] = lori[l_row<SM>(sm.li[t])]; sm.y.cc[t] = rori[r_pt<SM>(sm.ri[t])]; }
        for (int i = lane; i < num * W; i += 64) sm.hb[i / W][i % W] = 0u;
    } else {
        for (int t = lane; t < num; t += 64) { sm.x.s.lo[t] = lori[l_row<SM>(sm.li[t])]; sm.x.s.ro[t] = rori[r_pt<SM>(sm.ri[t])]; }
        for (int i = lane; i < num * W; i += 64) sm.hb[i / W][i % W] = 0u;
        const float s0 = (float)(1.0 / num);                                   // :1558
        for (int t = lane; t < SM::N4; t += 64) { sm.b[t] = t < num ? s0 : 0.0f; sm.y.cc[t] = 0.0f; }
    }
    WSYNC();
    // row t visits the pairs (t, t+d mod num), d = 1..num/2: every unordered pair once, evaluated as (lower, higher) index.
    // Survivor lists are short (usually < 32): the lanes of a partial block of <= 32 rows are split into groups that take different
    // offsets d of the same rows, as in dist_filter.
    const int half = num >> 1;
    const bool even = !(num & 1);
    auto pair = [&](int t, int k) {
        const int i = t < k ? t : k, j = t < k ? k : t;
        bool ok;
        if constexpr (SM::ALIAS_ORI) ok = angle_compatible(unpack_xy(sm.xy[i]), sm.b[i], sm.y.cc[i], unpack_xy(sm.xy[j]), sm.b[j], sm.y.cc[j]);
        else ok = angle_compatible(unpack_xy(sm.xy[i]), sm.x.s.lo[i], sm.x.s.ro[i], unpack_xy(sm.xy[j]), sm.x.s.lo[j], sm.x.s.ro[j]);
        if (ok) {
            atomicOr(&sm.hb[i][j >> 5], 1u << (j & 31));
            atomicOr(&sm.hb[j][i >> 5], 1u << (i & 31));
        }
    };
    const int tail0 = num & ~63, R = num - tail0;
    const bool grouped = R > 0 && R <= 32;
    const int n_wide = grouped ? tail0 : num;
    for (int d = 1; n_wide > 0 && d <= half; ++d) {
        for (int t = lane; t < n_wide; t += 64) {
            if (d == half && even && t >= half) continue;
            int k = t + d; if (k >= num) k -= num;
            pair(t, k);
        }
    }
    if (grouped) {
        const int sh = R <= 8 ? 3 : R <= 16 ? 4 : 5, RG = 1 << sh, dstep = 64 >> sh;
        const int r = lane & (RG - 1), t = tail0 + r;
        int d = 1 + (lane >> sh);
        int k = t + d; k = k >= num ? k - num : k;
        for (int d0 = 1; d0 <= half; d0 += dstep) {
            if (r < R && d <= half && !(d == half && even && t >= half)) pair(t, k);
            d += dstep; k += dstep; k = k >= num ? k - num : k;
        }
    }
    WSYNC();
    if constexpr (SM::ALIAS_ORI) {
        const float s0 = (float)(1.0 / num);                               // :1558
        for (int t = lane; t < SM::N4; t += 64) { sm.b[t] = t < num ? s0 : 0.0f; sm.y.cc[t] = 0.0f; }
        WSYNC();
    }
    GPH(PH + 4);
    for (int it = 0; it < 5; ++it) {                                       // :1563-1581
        for (int t = lane; t < num; t += 64) {
            float s1 = 0.0f;
            for (int w = 0; w < (num + 31) / 32; ++w) {
                uint32_t bits = sm.hb[t][w];
                while (bits) { const int k = w * 32 + __ffs(bits) - 1; bits &= bits - 1; s1 += sm.b[k]; }
            }
            sm.y.cc[t] = s1;
        }
        WSYNC();
        float sum = seq_sum(sm, num);
        sum = (float)(1.0 / ((double)sum + 0.00001));
        for (int t = lane; t < num; t += 64) sm.b[t] = sm.y.cc[t] * sum;
        WSYNC();
    }
    GPH(PH + 5);
    sort_scores<SM, REF_TIE>(sm, num, 0.001);
    const int nsel = greedy(sm, num, 0.001, [&sm](int a, int o) { return (sm.hb[a][o >> 5] >> (o & 31)) & 1u; });
    compact(sm, nsel);
    GPH(PH + 6);
    return nsel;
}

// both graph stages + the final sum; a list of fewer than 2 correspondences cannot survive S9 (a single node ends with S = 0)
template <class SM, bool LOOKUP, int ITERS, int REF_TIE>
__device__ __forceinline__ float graph_score(SM& sm, int num, const float* __restrict__ table, const float* __restrict__ lori,
                                             const float* __restrict__ rori, const float* __restrict__ ext, int& n_survivors, int stop_after = 2, int mode = 0)
{
    n_survivors = num;                                                     // stop_after 0: the candidate list itself (S3 / S7)
    if (stop_after == 0) return 0.0f;
    // instantiations rather than flags inside the loops: the register budget is that of the path taken
    if (mode == 2 && LOOKUP) num = dist_filter<SM, LOOKUP, ITERS, 2, REF_TIE>(sm, num, table, ext);
    else if (mode >= 1) num = dist_filter<SM, LOOKUP, ITERS, 1, REF_TIE>(sm, num, table, ext);
    else num = dist_filter<SM, LOOKUP, ITERS, 0, REF_TIE>(sm, num, table, ext);
    n_survivors = num;                                                     // stop_after 1: corr2, the survivors of S8
    if (mode >= 1) {                                                       // the survivors' points back to 16-bit integers: what S9 and the correspondence export read
        for (int t = threadIdx.x; t < num; t += 64) {
            const int2 v = sm.xy[t];
            int lx, ly, rx, ry;
            unpack_h2(v.x, lx, ly); unpack_h2(v.y, rx, ry);
            sm.xy[t] = pack_xy(lx, ly, rx, ry);
        }
        WSYNC();
    }
    if (stop_after == 1) return 0.0f;
    n_survivors = 0;
    if (num < 2) return 0.0f;
    num = angle_filter<SM, REF_TIE>(sm, num, lori, rori);
    n_survivors = num;                                                     // li/ri/xy[0..num) (+ similarities) = corr3 in the reference's order
    // :508-514 / :775-781: the sum of the survivors' similarities in list order.  They are gathered into b[] first (one parallel
    // round trip when they come from global memory) and added up sequentially from LDS.
    for (int t = threadIdx.x; t < num; t += 64) sm.b[t] = list_sim(sm, ext, t);
    WSYNC();
    float score = 0.0f;
    for (int i = 0; i < num; ++i) score += sm.b[i];
    return score;
}

// parity tap (tests only): the list a task holds after stage `stage` (0 = candidates, 1 = after S8, 2 = after S9)
struct GraphTap { MinuCand* out; int32_t* n; int stage; };
template <class SM>
__device__ __forceinline__ void tap_write(const GraphTap& tap, const SM& sm, const float* __restrict__ ext, long long task, int n, int cap)
{
    for (int t = threadIdx.x; t < n; t += 64) { MinuCand c; c.sim = list_sim(sm, ext, t); c.li = l_row<SM>(sm.li[t]); c.ri = r_pt<SM>(sm.ri[t]); tap.out[(size_t)task * cap + t] = c; }
    if (threadIdx.x == 0) tap.n[task] = n;
}

// =====================================================================================================================
// texture lists: S7 (top-200 rows of the ADC row maxima) + S8b + S9
// =====================================================================================================================
// 14 lists per CU: 11 200 B of LDS (8 neighbour indices per row; sort keys, picked rows and the angle stage's orientations share what is left) and 128 registers
// (26 spilled).  Measured (texture stage, 100k templates): 16 indices at 12 lists per CU 47.6 ms, 8 indices at 12 lists 51.6, 8 indices at 14 lists 46.2.
#ifndef AFIS_TEX_CACHE
#define AFIS_TEX_CACHE 2
#endif
#ifndef AFIS_TEX_WAVES
#define AFIS_TEX_WAVES 4
#endif
typedef WaveSmem<kTopTex, AFIS_TEX_CACHE, false> TexSmem;
constexpr int kTexRegs = (kTexMax + 63) / 64;     // 16 row maxima per lane: the wave holds all <= 1000 keys in registers

template <int REF_TIE>   // 1: option ref_tie_order 2 (sort_scores); its own instantiation, so that the default kernel is the code it was
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AFIS_TEX_WAVES, AFIS_TEX_WAVES))) void k_graph_texture(QueryDev q, GalleryDev g, const float* __restrict__ table_dist,
                                                      const float* __restrict__ rm_val, const int32_t* __restrict__ rm_arg,
                                                      const float* __restrict__ rm_cv, const int32_t* __restrict__ rm_n,
                                                      float* __restrict__ parts, GraphTap tap)
{
    __shared__ TexSmem sm;
    const int lane = threadIdx.x;
    GPH_ZERO();
    const int n_tasks = q.nq * g.G;
    for (;;) {
        const int task = next_task(g.task_ctr + 0);                       // lists differ widely in cost (a mate's is 10 x a non-mate's): tasks are drawn, not dealt
        if (task >= n_tasks) break;
        GPH_T0();
        const int qi = task / g.G, gi = task - qi * g.G;
        const int l0 = q.lt_off[qi], n_lt = q.lt_off[qi + 1] - l0;
        const int r0 = g.tex_off[gi], n_rt = g.tex_off[gi + 1] - r0;
        float* out = parts + (size_t)task * 4 + 3;
        if (n_lt <= 0 || n_rt <= 0) { if (lane == 0) { *out = 0.0f; if (tap.out) tap.n[task] = -1; } continue; }   // matcher.cpp:411: scorer not called
        const size_t o = (size_t)task * q.lt_pad;
        int num;
        if (n_lt > kTopTex) {                                            // :736-747: the 200 rows with the largest maxima
            // compact form (adc_variant 9): the rows that can be among the 200 — a third of them — side by side in row order as (value, row | point << 16);
            // otherwise every row's value and point at its row, -inf for rows that cannot
            const bool compact = rm_n != nullptr;
            const int n_in = compact ? rm_n[task] : n_lt;                // >= 200: the selection keeps at least the 200 rows with the largest lower bounds
            const int n_regs = (n_in + 63) >> 6;                         // registers per lane that hold a row (uniform)
            uint32_t key[kTexRegs]; int arg[kTexRegs];
#pragma unroll
            for (int u = 0; u < kTexRegs; ++u) {
                const int e = u * 64 + lane;
                const bool in = e < n_in;
                key[u] = 0u; arg[u] = 0;                                 // real keys are never 0
                if (u < n_regs) {                                        // uniform
                    key[u] = in ? g_ord_f32(compact ? rm_cv[o + e] : rm_val[o + e]) : 0u;
                    arg[u] = in ? rm_arg[o + e] : 0;                     // fetched with the values: one round trip, not one per picked row
                }
            }
            uint32_t T = 0;                                              // 200th largest key, built bit by bit ...
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = T | (1u << bit);
                int c = 0;
#pragma unroll
                for (int u = 0; u < kTexRegs; ++u) if (u < n_regs) c += g_wave_popc(key[u] >= cand);
                if (c >= kTopTex) { T = cand; if (c == kTopTex) break; } // ... or until exactly 200 keys are >= the prefix: they are the set
            }
            int n_gt = 0;
#pragma unroll
            for (int u = 0; u < kTexRegs; ++u) if (u < n_regs) n_gt += g_wave_popc(key[u] > T);
            const int need = kTopTex - n_gt;                             // of the keys equal to T keep the lowest indices
            int base_gt = 0, base_eq = 0;
            uint32_t* const key32 = sm.x.pick.keys;   // 200 ordered-float keys, read four at a time below
#pragma unroll
            for (int u = 0; u < kTexRegs; ++u) {                         // u ascending, lane ascending = index ascending
                if (u < n_regs) {
                    const int e = u * 64 + lane;
                    const bool gt = key[u] > T, eq = key[u] == T;
                    const u64 mg = __ballot(gt), me = __ballot(eq);
                    int pos = -1;
                    if (gt) pos = base_gt + g_lane_prefix(mg);
                    else if (eq) { const int r = base_eq + g_lane_prefix(me); if (r < need) pos = n_gt + r; }
                    if (pos >= 0) { key32[pos] = key[u]; sm.x.pick.te[pos] = tex_pack_l(compact ? arg[u] & 0xffff : e, e); sm.x.pick.targ[pos] = tex_pack_r(compact ? arg[u] >> 16 : arg[u], e); }   // e = the row's slot in the array the values came from
                    base_gt += __popcll(mg); base_eq += __popcll(me);
                }
            }
            WSYNC();
            num = kTopTex;
            // rank by counting on the 32-bit keys.  Picked rows sit in index order, so equal keys would need the index as a tie-break:
            // ties make the ranks collide, which their sum shows (a permutation of 0..199 sums to 19900, anything else to less); the
            // 64-bit (key, ~index) composites are ranked only then.
            // The 200 keys are first dealt into 64 bins of equal width between the smallest and the largest (counting sort through LDS counters; b[] and
            // cc[] are free until the distance stage), highest bin first; a key then counts the larger keys of ITS bin only and adds the bins above it:
            // 200 x 200 comparisons become 200 x (a bin's population).
            uint32_t m32[TexSmem::U]; int r[TexSmem::U];
            uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(sm.b);          // [64] fill pointers, [64..128) bin starts
            uint32_t* const s_gkey = reinterpret_cast<uint32_t*>(sm.y.cc);      // [200] the keys grouped by bin
            int dmax = 0;
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) {
                const int t = lane + 64 * u;
                m32[u] = t < num ? key32[t] : T;
                dmax = max(dmax, (int)((m32[u] - T) >> 1));                       // keys are >= T (the 200th largest): the difference is non-negative — but NOT small when the 200 row maxima have both signs
            }                                                                     // (ordered keys of +3 and -3 are 2^31 + 2^23 apart): HALF of it is what fits an int.  Rounds 3-5 took the difference itself: negative
            dmax = g_wave_max(dmax);                                              // for such lists, a zero shift, bins far beyond 63, counters scattered over the list's LDS and ranking loops of 2^31 trips (72 s per
            const int shift = max(0, 27 - __clz(dmax | 1));                       // search of a 224-row latent against structured prints; results still right: the tie fallback re-ranks).  bin = (key - T) >> shift in [0, 63]
            int bin[TexSmem::U];
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) bin[u] = min((int)((m32[u] - T) >> shift), 63);
            s_cnt[lane] = 0u;
            WSYNC();
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) if (lane + 64 * u < num) atomicAdd(&s_cnt[bin[u]], 1u);
            WSYNC();
            {   // lane l owns bin l: start = number of keys in the bins above it
                const int own = (int)s_cnt[lane];
                int suf = own;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_down(suf, off); if (lane + off < 64) suf += v; }
                WSYNC();
                s_cnt[lane] = (uint32_t)(suf - own); s_cnt[64 + lane] = (uint32_t)(suf - own);
            }
            WSYNC();
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) if (lane + 64 * u < num) s_gkey[atomicAdd(&s_cnt[bin[u]], 1u)] = m32[u];
            WSYNC();
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) {
                r[u] = 0;
                if (lane + 64 * u < num) {
                    const int lo = (int)s_cnt[64 + bin[u]], hi = (int)s_cnt[bin[u]];   // the fill pointer ended at the bin's end
                    int c = lo;
                    for (int k = lo; k < hi; ++k) c += s_gkey[k] > m32[u];
                    r[u] = c;
                }
            }
            int rsum = 0;
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) rsum += r[u];            // lanes beyond the list hold the maximum key: rank 0
            if (g_wave_sum(rsum) != kTopTex * (kTopTex - 1) / 2) {
                u64 mine[TexSmem::U];
#pragma unroll
                for (int u = 0; u < TexSmem::U; ++u) { const int t = lane + 64 * u; mine[u] = t < num ? ((u64)key32[t] << 32) | (uint32_t)(~(uint32_t)(sm.x.pick.te[t] & 1023)) : 0ull; r[u] = 0; }
                for (int k = 0; k < num; ++k) {
                    const u64 kk = ((u64)key32[k] << 32) | (uint32_t)(~(uint32_t)(sm.x.pick.te[k] & 1023));
#pragma unroll
                    for (int u = 0; u < TexSmem::U; ++u) r[u] += kk > mine[u];
                }
            }
#pragma unroll
            for (int u = 0; u < TexSmem::U; ++u) {
                const int t = lane + 64 * u;
                if (t < num) { sm.li[r[u]] = sm.x.pick.te[t]; sm.ri[r[u]] = sm.x.pick.targ[t]; }
            }
        } else {                                                         // :748-749 rows stay in index order
            num = n_lt;
            for (int t = lane; t < num; t += 64) { sm.li[t] = tex_pack_l(t, t); sm.ri[t] = tex_pack_r(rm_n ? rm_arg[o + t] >> 16 : rm_arg[o + t], t); }   // compact form: every row is listed, slot = row
        }
        WSYNC();
        int out_of_range = 0, not_small = 0;                             // any block coordinate outside [0, 2047]: generic arithmetic for this list;
        short2 lp4[TexSmem::U], rp4[TexSmem::U];                         // all inside [0, 49] (always, for real templates): no |d| < 50 test needed
#pragma unroll
        for (int u = 0; u < TexSmem::U; ++u) {
            const int t = lane + 64 * u;
            lp4[u] = make_short2(0, 0); rp4[u] = make_short2(0, 0);
            if (t < num) {
                lp4[u] = q.lt_xy[l0 + (sm.li[t] & 1023)]; rp4[u] = g.tex_xy[r0 + (sm.ri[t] & 1023)];
                out_of_range |= (lp4[u].x | lp4[u].y | rp4[u].x | rp4[u].y) & ~2047;
                not_small |= ((unsigned)lp4[u].x > 49u) | ((unsigned)lp4[u].y > 49u) | ((unsigned)rp4[u].x > 49u) | ((unsigned)rp4[u].y > 49u);
            }
        }
        const int mode = __ballot(out_of_range != 0) != 0ull ? 0 : (__ballot(not_small != 0) == 0ull ? 2 : 1);
#pragma unroll
        for (int u = 0; u < TexSmem::U; ++u) {                           // packed paths: coordinates as fp16 pairs (graph_arith.h); generic: 16-bit integers
            const int t = lane + 64 * u;
            if (t < num) sm.xy[t] = mode ? make_int2(pack_h2(lp4[u].x, lp4[u].y), pack_h2(rp4[u].x, rp4[u].y)) : pack_xy(lp4[u].x, lp4[u].y, rp4[u].x, rp4[u].y);
        }
        WSYNC();
        GPH_K(15);                                                       // S7 + list build
        int n_surv;
        const float* const row_max = (rm_n ? rm_cv : rm_val) + o;           // the array the entries' slots index
        const float score = graph_score<TexSmem, true, 3, REF_TIE>(sm, num, table_dist, q.lt_ori + l0, g.tex_ori + r0, row_max, n_surv, tap.out ? tap.stage : 2, mode);   // :759, :767
        if (lane == 0) *out = score;
        if (tap.out) tap_write(tap, sm, row_max, task, n_surv, kTopTex);
        WSYNC();
    }
    GPH_FLUSH();
}

hipError_t launch_graph_texture(const QueryDev& q, const GalleryDev& g, const float* table_dist,
                                const float* rm_val, const int32_t* rm_arg, const float* rm_cv, const int32_t* rm_n, float* parts, MinuCand* tap_out, int32_t* tap_n, int tap_stage, hipStream_t stream)
{
    const long long n_tasks = (long long)q.nq * g.G;
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7ffffff0LL || !g.task_ctr) return hipErrorInvalidValue;
    const int grid = (int)(n_tasks < 16384 ? n_tasks : 16384);
    hipError_t e0 = hipMemsetAsync(g.task_ctr + 0, 0, 4, stream);
    if (e0 != hipSuccess) return e0;
    const GraphTap tap{tap_out, tap_n, tap_stage & 255};
    if ((tap_stage >> 8) & 1) hipLaunchKernelGGL(k_graph_texture<1>, dim3(grid), dim3(64), 0, stream, q, g, table_dist, rm_val, rm_arg, rm_cv, rm_n, parts, tap);
    else hipLaunchKernelGGL(k_graph_texture<0>, dim3(grid), dim3(64), 0, stream, q, g, table_dist, rm_val, rm_arg, rm_cv, rm_n, parts, tap);
    return hipGetLastError();
}

// =====================================================================================================================
// minutiae lists (produced by k_minu_cands, already in rank order): S8a + S9
// =====================================================================================================================
#ifndef AFIS_MINU_CACHE
#define AFIS_MINU_CACHE 3
#endif
typedef WaveSmem<kTopMinu, AFIS_MINU_CACHE, true> MinuGraphSmem;

// corr_out / corr_n (optional): the surviving correspondences of every task as (lx, ly, rx, ry), matcher.cpp:497-505
// Six lists per SIMD: 6240 B of LDS (12 neighbour indices per row, no value stash: recomputing a value costs 20 instructions, the stash cost waves; the angle
// stage's orientations borrow b[] / cc[]) and 80 registers.  Measured at 100k templates (minutiae stage, candidates included): 4 waves with 6 values + 16 indices per row
// 93.7 ms, 5 waves with 24 indices 89.7-90.4, 6 waves with 12 indices 88.9 (12 indices at 5 waves: 92.3).
#ifndef AFIS_MINU_WAVES
#define AFIS_MINU_WAVES 6
#endif
template <int REF_TIE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AFIS_MINU_WAVES, AFIS_MINU_WAVES))) void k_graph_minutiae(QueryDev q, GalleryDev g, const MinuCand* __restrict__ cands,
                                                       const int32_t* __restrict__ cand_n, float* __restrict__ parts,
                                                       short4* __restrict__ corr_out, int32_t* __restrict__ corr_n, GraphTap tap)
{
    __shared__ MinuGraphSmem sm;
    const int lane = threadIdx.x;
    GPH_ZERO();
    const int n_tasks = q.nq * 3 * g.G;
    for (;;) {
        const int task = next_task(g.task_ctr + 1);
        if (task >= n_tasks) break;
        GPH_T0();
        // task order as in k_minu_cands: gallery template fastest, then selected template, then query
        const int gi = task % g.G;
        const int qs = task / g.G;
        const int qi = qs / 3, s = qs - qi * 3;
        float* out = parts + ((size_t)qi * g.G + gi) * 4 + s;
        const int num = cand_n[task];
        if (num <= 0) { if (lane == 0) { *out = 0.0f; if (corr_n) corr_n[task] = 0; if (tap.out) tap.n[task] = -1; } continue; }   // matcher.cpp:400-404
        const int l0 = q.lm_off[qs], r0 = g.minu_off[gi];
        const MinuCand* c = cands + (size_t)task * kTopMinu;
        int out_of_range = 0;                                            // any pixel coordinate outside [0, 2047]: generic float arithmetic
        short2 lp2[MinuGraphSmem::U], rp2[MinuGraphSmem::U];
#pragma unroll
        for (int u = 0; u < MinuGraphSmem::U; ++u) {
            const int t = lane + 64 * u;
            lp2[u] = make_short2(0, 0); rp2[u] = make_short2(0, 0);
            if (t < num) {
                const MinuCand cd = c[t];
                sm.simv[t] = cd.sim; sm.li[t] = cd.li; sm.ri[t] = cd.ri;
                lp2[u] = q.lm_xy[l0 + cd.li]; rp2[u] = g.minu_xy[r0 + cd.ri];
                out_of_range |= (lp2[u].x | lp2[u].y | rp2[u].x | rp2[u].y) & ~2047;
            }
        }
        const int mode = __ballot(out_of_range != 0) == 0ull ? 1 : 0;
#pragma unroll
        for (int u = 0; u < MinuGraphSmem::U; ++u) {                     // packed path: coordinates as fp16 pairs (graph_arith.h); generic: 16-bit integers
            const int t = lane + 64 * u;
            if (t < num) sm.xy[t] = mode ? make_int2(pack_h2(lp2[u].x, lp2[u].y), pack_h2(rp2[u].x, rp2[u].y)) : pack_xy(lp2[u].x, lp2[u].y, rp2[u].x, rp2[u].y);
        }
        WSYNC();
        GPH_K(7);                                                        // list load
        int n_surv;
        const float score = graph_score<MinuGraphSmem, false, 5, REF_TIE>(sm, num, nullptr, q.lm_ori + l0, g.minu_ori + r0, nullptr, n_surv, tap.out ? tap.stage : 2, mode);   // :492, :495
        if (lane == 0) *out = score;
        if (tap.out) tap_write(tap, sm, nullptr, task, n_surv, kTopMinu);
        if (corr_out) {
            for (int t = lane; t < n_surv; t += 64) {
                const Pt p = unpack_xy(sm.xy[t]);
                corr_out[(size_t)task * kTopMinu + t] = make_short4((short)p.lx, (short)p.ly, (short)p.rx, (short)p.ry);
            }
            if (lane == 0) corr_n[task] = n_surv;
        }
        WSYNC();
    }
    GPH_FLUSH();
}

hipError_t launch_graph_minutiae(const QueryDev& q, const GalleryDev& g, const MinuCand* cands, const int32_t* cand_n,
                                 float* parts, short4* corr_out, int32_t* corr_n, MinuCand* tap_out, int32_t* tap_n, int tap_stage, hipStream_t stream, bool join)
{
    // join: a second instance of the kernel on another stream that draws from the SAME list counter as one already running (afis_search.cpp, option bound_cus): no reset
    const long long n_tasks = (long long)q.nq * 3 * g.G;
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7ffffff0LL || !g.task_ctr) return hipErrorInvalidValue;
    const int grid = (int)(n_tasks < 32768 ? n_tasks : 32768);
    hipError_t e0 = join ? hipSuccess : hipMemsetAsync(g.task_ctr + 1, 0, 4, stream);
    if (e0 != hipSuccess) return e0;
    const GraphTap tap{tap_out, tap_n, tap_stage & 255};
    if ((tap_stage >> 8) & 1) hipLaunchKernelGGL(k_graph_minutiae<1>, dim3(grid), dim3(64), 0, stream, q, g, cands, cand_n, parts, corr_out, corr_n, tap);
    else hipLaunchKernelGGL(k_graph_minutiae<0>, dim3(grid), dim3(64), 0, stream, q, g, cands, cand_n, parts, corr_out, corr_n, tap);
    return hipGetLastError();
}

// parity tap: the device's atan2 on the grid of integer coordinate differences, out[(dy + R) * (2R + 1) + (dx + R)]
__global__ __launch_bounds__(256) void k_debug_atan2_grid(int R, float* __restrict__ out)
{
    const int W = 2 * R + 1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)W * W) return;
    const int dy = (int)(idx / W) - R, dx = (int)(idx % W) - R;
    out[idx] = atan2_f32((float)dy, (float)dx);
}
hipError_t launch_debug_atan2_grid(int R, float* out, hipStream_t stream)
{
    const long long n = (long long)(2 * R + 1) * (2 * R + 1);
    hipLaunchKernelGGL(k_debug_atan2_grid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, R, out);
    return hipGetLastError();
}

// parity tap: the packed paths' arithmetic (graph_arith.h) against the straightforward evaluation it replaces.
//   part 0: sqrt_rn_int(n) vs sqrt_rn_pos(n), every integer n in [0, 2 * 2047^2]                      -> cnt[0] = mismatches
//   part 1: texture "H != 0", every (n1, n2) in [0, 4802]^2                                            -> cnt[1..3] = pairs, inside the band, wrong outside it
//   part 2: minutiae "H != 0", n1 over [0, 2 * 2047^2], n2 within +-12 of (sqrt n1 +- 30)^2            -> cnt[4..6]
//   part 3: diff_n / pack_h2 / unpack_h2 on every coordinate difference of [-2047, 2047]^2                 -> cnt[7] = mismatches
__device__ __forceinline__ void arith_check(bool tex, float f1, float f2, unsigned long long* cnt)
{
    const float d = fabsf(sqrt_rn_pos(f1) - sqrt_rn_pos(f2));
    const bool ref = tex ? 16.0f * d < 30.0f : d < 30.0f;
    const int a = tex ? pair_compatible_alg<true>(f1, f2) : pair_compatible_alg<false>(f1, f2);
    const bool full = tex ? pair_compatible_n<true>(f1, f2) : pair_compatible_n<false>(f1, f2);
    atomicAdd(cnt + 0, 1ull);
    if (a == 2) atomicAdd(cnt + 1, 1ull);
    if ((a != 2 && (a == 1) != ref) || full != ref) atomicAdd(cnt + 2, 1ull);
}
__global__ __launch_bounds__(256) void k_debug_graph_arith(int part, unsigned long long* __restrict__ cnt)
{
    constexpr unsigned kMaxN = 2u * 2047u * 2047u;
    if (part == 0) {
        const unsigned n = blockIdx.x * 256u + threadIdx.x;
        if (n <= kMaxN && __float_as_uint(sqrt_rn_int((float)n)) != __float_as_uint(sqrt_rn_pos((float)n))) atomicAdd(cnt, 1ull);
    } else if (part == 3) {                                               // diff_n on fp16-packed points vs the integer dx^2 + dy^2, every (dx, dy) of [-2047, 2047]^2
        const int dx = (int)blockIdx.x - 2047, dy = (int)(blockIdx.y * 256u + threadIdx.x) - 2047;
        if (dy > 2047) return;
        const int ax = dx > 0 ? dx : 0, ox = dx > 0 ? 0 : -dx, ay = dy > 0 ? dy : 0, oy = dy > 0 ? 0 : -dy;
        const float n = diff_n(pack_h2(ax, ay), pack_h2(ox, oy));
        int bx, by; unpack_h2(pack_h2(ax, ay), bx, by);
        if (__float_as_uint(n) != __float_as_uint((float)(dx * dx + dy * dy)) || bx != ax || by != ay) atomicAdd(cnt, 1ull);
    } else if (part == 1) {
        const unsigned n1 = blockIdx.x, n2 = blockIdx.y * 256u + threadIdx.x;
        if (n1 <= 4802u && n2 <= 4802u) arith_check(true, (float)n1, (float)n2, cnt + 1);
    } else {
        const unsigned n1 = blockIdx.x * 256u + threadIdx.x;
        if (n1 > kMaxN) return;
        const double a = sqrt((double)n1);
        for (int sg = -1; sg <= 1; sg += 2) {
            const double b = a + 30.0 * sg;
            if (b < 0) continue;
            const long long c = (long long)(b * b + 0.5);
            for (int j = -12; j <= 12; ++j) { const long long n2 = c + j; if (n2 >= 0 && n2 <= (long long)kMaxN) arith_check(false, (float)n1, (float)n2, cnt + 4); }
        }
    }
}
hipError_t launch_debug_graph_arith(unsigned long long* cnt8, hipStream_t stream)
{
    constexpr unsigned kMaxN = 2u * 2047u * 2047u;
    hipError_t e = hipMemsetAsync(cnt8, 0, 64, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_debug_graph_arith, dim3((kMaxN + 256) / 256), dim3(256), 0, stream, 0, cnt8);
    hipLaunchKernelGGL(k_debug_graph_arith, dim3(4803, 19), dim3(256), 0, stream, 1, cnt8);
    hipLaunchKernelGGL(k_debug_graph_arith, dim3((kMaxN + 256) / 256), dim3(256), 0, stream, 2, cnt8);
    hipLaunchKernelGGL(k_debug_graph_arith, dim3(4095, 16), dim3(256), 0, stream, 3, cnt8 + 7);
    return hipGetLastError();
}

hipError_t read_graph_phase_cycles(unsigned long long* out16, bool reset)
{
#ifdef AFIS_PHASE_TIMING
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_graph_phase), 16 * sizeof(u64));
    if (e != hipSuccess) return e;
    if (reset) { u64 z[16] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_graph_phase), z, sizeof(z)); }
    return e;
#else
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    return hipSuccess;
#endif
}

}  // namespace afis
