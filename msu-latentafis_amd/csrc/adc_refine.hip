// adc_refine.hip — the parts of adc_variant 9 around the matrix-core bound pass (adc_mfma.hip, which see for the method):
//   k_mf_codebook / k_mf_tiles  the fp16 codebook and the tile-aligned code stream (once per context / gallery)
//   k_mf_rows                   per latent texture row: fp16 operand fragments and the rigorous error constants (c_i, Es_i, Tg_i, force)
//   k_tex_refine                per (latent, rolled) pair: the rows that can reach the top 200 (S7, matcher.cpp:736-747) by their bounds, and for those
//                               the reference's own table arithmetic (include.h:327-359, matcher.cpp:571-592) on the candidate cells.
// Built WITHOUT -fno-honor-nans (unlike adc_mfma.hip): forced rows carry Tg = Es = inf, a NaN / inf latent descriptor makes exact_sim NaN / -inf, and the
// comparisons these kernels make on such values must keep their IEEE meaning.
#include "afis_device.h"
#include <algorithm>

namespace afis {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr float kMfNeg = -1.0e30f;          // "no point": the accumulator start of padding points

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __uint_as_float(x); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // v_max3_f32
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// ---------------------------------------------------------------------------------------------------------------------------------
// Codebook in fp16, one 16-byte entry per (m, c): 6 halves (round to nearest) + 4 bytes of padding; and |cw_mc|^2 per entry.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mf_codebook(const float* __restrict__ cw, uint4* __restrict__ cw16, float* __restrict__ cwn)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float* w = cw + (size_t)e * kDsub;
    uint32_t h[6];
    double n2 = 0.0;
#pragma unroll
    for (int d = 0; d < kDsub; ++d) {
        const _Float16 x = (_Float16)w[d];
        h[d] = (uint32_t)__builtin_bit_cast(unsigned short, x);
        n2 += (double)w[d] * (double)w[d];
    }
    cw16[e] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), 0u);
    cwn[e] = (float)n2;
}

// Tile-aligned copy of the gallery's texture codes: template t owns ceil(n/32) tiles (32 entries of 16 code bytes, zero beyond the template's
// points) starting at tile t_blk[t]; per entry also G's point term -|b_j|^2 / 2 (kMfNeg beyond the points), and per tile
// (template, tile index in the template | 256 on the template's last tile).  grid = G, block = 64.
//
// Repeated code vectors.  Texture descriptors of neighbouring grid points come from overlapping patches (extraction_rolled.py:112-141) and are PQ-encoded
// afterwards: in smooth regions several points of a template carry the SAME 16 code bytes.  Such points have the same similarity to every latent row, bit for bit
// (the same table entries in the same order, matcher.cpp:571-592), and std::max_element (matcher.cpp:730) takes the first: a later occurrence can never be a row's
// arg-max and never changes a row's maximum.  So only the FIRST occurrence of a code vector takes part in the bound pass — the others get the point term of a
// padding point (kMfNeg) and zero codes.  Without this a row whose best vector occurs three times has three cells within any tolerance of its maximum, the bound
// pass gives up on it ("many") and the recomputation evaluates every point of the template for it: at 30 % repeated points that was 4 % of the evaluated rows and
// most of the recomputation's time (profiles/r06_bench_structured.json).  (A row evaluated in full still walks all points, repeated ones included: same result.)
__global__ __launch_bounds__(64) void k_mf_tiles(GalleryDev g, const int32_t* __restrict__ t_blk, const float* __restrict__ cwn,
                                                 uint4* __restrict__ codes_p, float* __restrict__ nrm_p, int2* __restrict__ tile_meta)
{
    __shared__ uint4 s_c[kTexMax];
    __shared__ uint32_t s_hash[kTexMax];
    const int t = blockIdx.x, lane = threadIdx.x;
    const int p0 = g.tex_off[t], n = min(g.tex_off[t + 1] - p0, kTexMax);
    const int nt = (n + 31) >> 5;
    for (int p = lane; p < n; p += 64) {
        const uint4 c = g.tex_codes[p0 + p];
        s_c[p] = c; s_hash[p] = (c.x * 0x9E3779B1u) ^ (c.y * 0x85EBCA77u) ^ (c.z * 0xC2B2AE3Du) ^ (c.w * 0x27D4EB2Fu);
    }
    __syncthreads();
    for (int p = lane; p < nt * 32; p += 64) {
        uint4 c = make_uint4(0, 0, 0, 0);
        float nrm = kMfNeg;
        if (p < n) {
            c = s_c[p];
            const uint32_t hp = s_hash[p];
            bool repeated = false;
            for (int q = 0; q < p && !repeated; ++q)                                // (the wave walks to its largest p: broadcast reads of s_hash[q]; the 16-byte compare only behind a hash match)
                if (s_hash[q] == hp) { const uint4 d = s_c[q]; repeated = d.x == c.x && d.y == c.y && d.z == c.z && d.w == c.w; }
            if (!repeated) {
                const uint32_t w[4] = {c.x, c.y, c.z, c.w};
                float s = 0.0f;
#pragma unroll
                for (int m = 0; m < kM; ++m) s += cwn[m * kK + ((w[m >> 2] >> (8 * (m & 3))) & 255u)];
                nrm = -0.5f * s;
            } else c = make_uint4(0, 0, 0, 0);
        }
        const size_t e = (size_t)t_blk[t] * 32 + p;
        codes_p[e] = c; nrm_p[e] = nrm;
        if ((p & 31) == 0) { const int k = p >> 5; tile_meta[t_blk[t] + k] = make_int2(t, k | (k == nt - 1 ? 256 : 0)); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Per latent texture row (grid = padded rows, block = 256 = codeword index): the row in fp16 as MFMA B-operand fragments, and its constants
//   rowk[i] = (c_i, Es_i, Tg_i, force)
// With a~ = fp16(a), b~ = fp16(cw), da = a - a~, db = cw - b~ (exact differences) and G = fl(a~ . b~) + n_j, n_j = fl(-|b_j|^2 / 2):
//   |G - (a . b_j - |b_j|^2 / 2)| <= Eg = sum_m max_c |da_m . cw_mc| + sum_m max_c |a~_m . db_mc| + accumulation + n_j rounding
// (per sub-quantizer the codeword of point j is ONE of the 256, so the per-m maxima bound every point), and
//   delta = the reference's own fp32 rounding of 6 - sum lut (19 roundings of intermediates <= 6 + sum_m max_c lut, as in adc.hip::k_lutq_build),
//   pert  = what replacing the low 6 mantissa bits of a tracked value by an index can move it.
//   Tg (units of G)   = 2 Eg + delta + 2 pert : a point whose REFERENCE similarity equals the row maximum has G >= G_best - Tg
//   Es (units of sim) = 2 Eg + delta + 2 pert + rounding of c_i : the reference's row maximum lies in c_i + 2 G_best +- Es.
// force = 1 (and Tg = Es = inf) for rows whose descriptors fp16 cannot carry (|a| > 1000, non-finite bounds): every cell is then evaluated exactly.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mf_rows(const float* __restrict__ lt_des, int n_rows, const float* __restrict__ cw, const float* __restrict__ cwn,
                                                 _Float16* __restrict__ bfrag, float4* __restrict__ rowk)
{
    __shared__ float s_a[kDes], s_ah[kDes];
    __shared__ float s_red[4][5];
    const int row = blockIdx.x, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const bool real = row < n_rows;
    if (c < kDes) {
        const float a = real ? lt_des[(size_t)row * kDes + c] : 0.0f;
        const bool fits = (f2u(a) & 0x7fffffffu) <= 0x447a0000u;        // |a| <= 1000 on the bits (false for NaN and inf)
        const _Float16 ah = (_Float16)(fits ? a : 0.0f);
        s_a[c] = a; s_ah[c] = (float)ah;
        const int kk = c >> 4, half = (c >> 3) & 1, e = c & 7, r = row & 31, rb = row >> 5;
        bfrag[(((size_t)rb * 6 + kk) * 64 + r + 32 * half) * 8 + e] = ah;
    }
    __syncthreads();
    if (!real) return;
    double P = 0.0, Q = 0.0, L = 0.0, S = 0.0, N = 0.0;
    for (int m = 0; m < kM; ++m) {
        const float* w = cw + ((size_t)m * kK + c) * kDsub;
        float p = 0.f, q = 0.f, sabs = 0.f, a6[kDsub], w6[kDsub];
#pragma unroll
        for (int d = 0; d < kDsub; ++d) {
            const float wf = w[d], wh = (float)(_Float16)wf, a = s_a[m * kDsub + d], ah = s_ah[m * kDsub + d];
            p = __builtin_fmaf(a - ah, wf, p); q = __builtin_fmaf(ah, wf - wh, q); sabs = __builtin_fmaf(fabsf(ah), fabsf(wh), sabs);
            a6[d] = a; w6[d] = wf;
        }
        float v[5] = {fabsf(p), fabsf(q), lut_entry(a6, w6), sabs, cwn[m * kK + c]};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float x = v[k];
            if ((f2u(x) & 0x7fffffffu) > 0x7f800000u) x = INFINITY;     // NaN input (tested on the bits: this file is built with -fno-honor-nans): unbounded
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
            v[k] = x;
        }
        __syncthreads();
        if (lane == 0) { for (int k = 0; k < 5; ++k) s_red[wave][k] = v[k]; }
        __syncthreads();
        if (c == 0) {
            float r[5];
            for (int k = 0; k < 5; ++k) r[k] = fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k]));
            P += r[0]; Q += r[1]; L += r[2]; S += r[3]; N += 0.5 * r[4];
        }
    }
    if (c == 0) {
        double A2 = 0.0; bool fits = true;
        for (int k = 0; k < kDes; ++k) { A2 += (double)s_a[k] * (double)s_a[k]; fits = fits && (f2u(s_a[k]) & 0x7fffffffu) <= 0x447a0000u; }
        const double u = 5.9604644775390625e-8;                          // 2^-24
        const double mag = S + N;                                        // no partial sum of the accumulation is larger
        const double Eg = 1.001 * (P + Q) + 100.0 * u * mag + 32.0 * u * N + 1e-9;
        const double pert = 128.0 * u * mag;                             // the low 6 mantissa bits of a tracked value carry an index
        const double delta = 24.0 * u * (6.0 + 1.0001 * L);
        double Tg = 2.0 * Eg + delta + 2.0 * pert;
        double Es = 2.0 * Eg + 2.0 * pert + delta + 4.0 * u * fmax(8.0, 6.0 + A2);
        const bool force = !fits || !(Tg < 1e20) || !(Es < 1e20) || (f2u((float)Tg) & 0x7fffffffu) > 0x7f800000u || (f2u((float)Es) & 0x7fffffffu) > 0x7f800000u;
        if (force) { Tg = INFINITY; Es = INFINITY; }
        rowk[row] = make_float4((float)(6.0 - A2), (float)(Es * 1.000001), (float)(Tg * 1.000001), force ? 1.0f : 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Selection by bounds + exact evaluation.  Persistent workgroups of 8 waves (the fp32 codebook, 96 KB, in LDS); each wave draws
// (latent, rolled) pairs from a counter and owns a 4 KB item list.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kRfWaves = 16, kRfItems = 384;        // 96 KB codebook + 16 x 3.75 KB item lists = 156 KB of LDS; <= 128 VGPRs
struct __align__(16) RfWave { unsigned short row[kRfItems]; unsigned short pt[kRfItems]; unsigned short slot[kRfItems]; float val[kRfItems]; };

__device__ __forceinline__ uint32_t ord_f32(float v) { const uint32_t b = f2u(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
// this wave's own LDS traffic in program order (the waves of the workgroup work on different pairs: no workgroup barrier may be used)
#define RF_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

__device__ __forceinline__ int rf_next_task(int32_t* ctr)               // one atomic by lane 0 (see graph.hip::next_task for why it is asm)
{
    int t;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(t), "=&s"(saved) : "v"(0), "v"(1), "s"(ctr) : "memory");
    return __builtin_amdgcn_readfirstlane(t);
}

__device__ __forceinline__ void rf_argmax(float& v, int& i)             // value descending, point ascending: the FIRST maximum
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        const bool take = (ov > v) | ((ov == v) & (oi < i));
        v = take ? ov : v; i = take ? oi : i;
    }
}

// stats (optional): [0] pairs, [1] rows, [2] active rows, [3] items evaluated, [4] rows evaluated in full, [5] rows whose exact maximum fell outside its bounds (must stay 0)
__global__ __launch_bounds__(kRfWaves * 64) void k_tex_refine(QueryDev q, GalleryDev g, const float* __restrict__ cw32, const uint2* __restrict__ rec,
                                                               const float4* __restrict__ rowk, int R_pad, int all_rows, float* __restrict__ rm_val,
                                                               int32_t* __restrict__ rm_arg, int32_t* __restrict__ task_ctr, unsigned long long* __restrict__ stats,
                                                               float* __restrict__ rm_cv, int32_t* __restrict__ rm_n)
{
    __shared__ float s_cw[kM * kK * kDsub];                             // 96 KB
    __shared__ RfWave s_w[kRfWaves];                                    // 32 KB
    for (int i = threadIdx.x; i < kM * kK * kDsub; i += kRfWaves * 64) s_cw[i] = cw32[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    RfWave& W = s_w[threadIdx.x >> 6];
    const int n_tasks = q.nq * g.G;
    for (;;) {
        const int task = rf_next_task(task_ctr);
        if (task >= n_tasks) break;
        const int qi = task / g.G, gi = task - qi * g.G;
        const int l0 = q.lt_off[qi], n_lt = q.lt_off[qi + 1] - l0;
        const int r0 = g.tex_off[gi], n_rt = g.tex_off[gi + 1] - r0;
        if (n_lt <= 0 || n_rt <= 0) continue;                           // no texture on one side: the scorer is not called (matcher.cpp:411)
        const size_t o = (size_t)task * q.lt_pad;
        const uint2* rec0 = rec + (size_t)gi * R_pad + l0;               // one record per (template, row): the bound pass merges the two lane halves (adc_mfma.hip)
        const float* des = q.lt_des + (size_t)l0 * kDes;

        // exact similarity of the latent row whose descriptor sits in W.val[0 .. 95] (rows evaluated over every point) and rolled point p: table entries recomputed (include.h:327-359), the four chains of matcher.cpp:571-592
        auto exact_sim_lds = [&](int p) -> float {
            const uint4 cd = g.tex_codes[r0 + p];
            const uint32_t w4[4] = {cd.x, cd.y, cd.z, cd.w};
            const float4* a4 = reinterpret_cast<const float4*>(W.val);
            float d[4] = {6.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
                __builtin_amdgcn_sched_barrier(0);
                float a[24];
#pragma unroll
                for (int k = 0; k < 6; ++k) { const float4 v = a4[mg * 6 + k]; a[4 * k] = v.x; a[4 * k + 1] = v.y; a[4 * k + 2] = v.z; a[4 * k + 3] = v.w; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int mm = 4 * mg + c;
                    const float2* wv = reinterpret_cast<const float2*>(s_cw + ((size_t)mm * kK + ((w4[mg] >> (8 * c)) & 255u)) * kDsub);
                    const float2 w0 = wv[0], w1 = wv[1], w2 = wv[2];
                    const float w6[kDsub] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
                    d[c] -= lut_entry(a + 6 * c, w6);
                }
            }
            return (d[0] + d[1]) + (d[2] + d[3]);
        };
        // ---- A: lower bounds of every row's maximum (ordered keys, 16 registers) ------------------------------------------------------------
        constexpr int kRegs = (kTexMax + 63) / 64;
        const int n_regs = (n_lt + 63) >> 6;
        auto bounds_rk = [&](const float4& rk, const uint2& a, float& lo, float& hi) {
            const float mid = rk.x + 2.0f * u2f(a.x);
            const float sl = rk.y + 4e-6f * fmaxf(1.0f, fabsf(mid));
            lo = mid - sl; hi = mid + sl;
            if (rk.w != 0.0f) { lo = -INFINITY; hi = INFINITY; }
        };
        auto bounds = [&](int e, const uint2& a, float& lo, float& hi) {
            const float4 rk = rowk[l0 + e];
            const float mid = rk.x + 2.0f * u2f(a.x);
            const float sl = rk.y + 4e-6f * fmaxf(1.0f, fabsf(mid));    // rounding of mid itself (|mid| <= a few units): two more ulps on either side
            lo = mid - sl; hi = mid + sl;
            if (rk.w != 0.0f) { lo = -INFINITY; hi = INFINITY; }        // a forced row (a descriptor fp16 cannot carry: rk.x may be -inf or NaN) is bounded by nothing
        };
        uint32_t C = 0u;                                                // rows whose UPPER bound's key is below C cannot be among the pair's top 200
        if (!all_rows && n_lt > kTopTex) {
            uint32_t klo[kRegs];
            uint32_t kmax = 0u, kmin = 0xffffffffu;
#pragma unroll
            for (int u = 0; u < kRegs; ++u) {
                const int e = u * 64 + lane;
                klo[u] = 0u;
                if (u < n_regs && e < n_lt) {
                    float lo, hi; bounds(e, rec0[e], lo, hi);
                    klo[u] = ord_f32(lo);
                    kmax = max(kmax, klo[u]); kmin = min(kmin, klo[u]);
                }
            }
            // ---- B: (a lower bound of) the 200th largest lower bound (matcher.cpp:736-747), bit by bit.  Any C at or below the true value is safe
            // (it only lets a few more rows through), so the search starts at the first bit in which the keys differ at all and stops 14 bits
            // further down: a resolution of 2^-14 of the keys' spread, far below the width of the bounds themselves.
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off)); kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off)); }
            const uint32_t diff = kmax ^ kmin;
            const int top = diff ? 31 - __clz((int)diff) : -1;           // highest differing bit (wave-uniform)
            C = top >= 0 ? (kmax & ~((2u << top) - 1u)) : kmax;          // the common prefix
            for (int bit = top; bit >= max(top - 14, 0); --bit) {
                const uint32_t cand = C | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < kRegs; ++u) if (u < n_regs) cnt += __popcll(__ballot(klo[u] >= cand));
                if (cnt >= kTopTex) C = cand;
            }
        }
        // ---- C: candidate items of the active rows ------------------------------------------------------------------------
        int n_items = 0, n_act = 0;
        const bool compact = rm_n != nullptr;
        unsigned long long st_active = 0, st_items = 0, st_full = 0;
        // The same value with FOUR lanes per item, lane = one of the reference's four chains (matcher.cpp:571-592: chain c subtracts the entries of sub-quantizers c, c + 4, c + 8, c + 12
        // in that order from 6 / 0 / 0 / 0; the chains meet as (d0 + d1) + (d2 + d3)): the four lanes of an item read 96 contiguous bytes of the latent row per step instead of every
        // lane walking its own 384-byte row (64 cache lines per load instruction, 24 instructions per item: the address path of the CU, not the arithmetic, was what the wave waited for).
        auto exact_sim4 = [&](int e, int p, int c) -> float {                // valid in the lanes with c == 0
            const uint4 cd = g.tex_codes[r0 + p];
            const uint32_t w4[4] = {cd.x, cd.y, cd.z, cd.w};
            const float2* a2 = reinterpret_cast<const float2*>(des + (size_t)e * kDes);
            float d = c == 0 ? 6.0f : 0.0f;
            float2 av[4][3];
#pragma unroll
            for (int mg = 0; mg < 4; ++mg)
#pragma unroll
                for (int k = 0; k < 3; ++k) av[mg][k] = a2[3 * (4 * mg + c) + k];
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
                const int mm = 4 * mg + c;
                const float2* wv = reinterpret_cast<const float2*>(s_cw + ((size_t)mm * kK + ((w4[mg] >> (8 * c)) & 255u)) * kDsub);
                const float2 w0 = wv[0], w1 = wv[1], w2 = wv[2];
                const float a6[kDsub] = {av[mg][0].x, av[mg][0].y, av[mg][1].x, av[mg][1].y, av[mg][2].x, av[mg][2].y};
                const float w6[kDsub] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
                d -= lut_entry(a6, w6);
            }
            const float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xF5, 0xf, 0xf, false));   // quad_perm [1, 1, 3, 3]: lane 0 <- d1, lane 2 <- d3
            const float s2 = d + t;                                                                                    // lane 0: d0 + d1, lane 2: d2 + d3
            const float u2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xAA, 0xf, 0xf, false)); // quad_perm [2, 2, 2, 2]
            return s2 + u2;                                                                                            // lane 0: (d0 + d1) + (d2 + d3)
        };
        auto flush = [&]() {
            RF_WSYNC();
            for (int it0 = 0; it0 < n_items; it0 += 16) {                   // sixteen items per trip
                const int it = it0 + (lane >> 2);
                const bool ok = it < n_items;
                const float v = exact_sim4(W.row[ok ? it : 0], W.pt[ok ? it : 0], lane & 3);
                if (ok && (lane & 3) == 0) W.val[it] = v;
            }
            RF_WSYNC();
            for (int it = lane; it < n_items; it += 64) {
                const int row = W.row[it];
                if (it == 0 || W.row[it - 1] != row) {                  // the first item of its row: reduce the row's run
                    float bv = W.val[it]; int bp = W.pt[it];
                    for (int j = it + 1; j < n_items && W.row[j] == row; ++j) {
                        const float v = W.val[j]; const int p = W.pt[j];
                        if (v > bv || (v == bv && p < bp)) { bv = v; bp = p; }
                    }
                    // compact form: ONE store of the value, in the pair's list of the rows that matter — S7 and the list's sums read it there through the row's slot (graph.hip);
                    // rounds 3-5 also stored it at its row of the dense array (a scattered 4-byte store per evaluated row: a third of this kernel's HBM writes)
                    if (compact) { const int sl = W.slot[it]; rm_cv[o + sl] = bv; rm_arg[o + sl] = row | (bp << 16); } else { rm_val[o + row] = bv; rm_arg[o + row] = bp; }
                    if (stats) {                                        // self-check of the bounds: the exact row maximum must lie inside them
                        float lo, hi; bounds(row, rec0[row], lo, hi);
                        if (!(bv >= lo && bv <= hi)) atomicAdd(stats + 5, 1ull);
                    }
                }
            }
            st_items += (unsigned long long)n_items;
            n_items = 0;
            RF_WSYNC();
        };
        // (the record and the row constants of round u + 1 are fetched before round u is worked on: one exposed memory round trip per pair instead of one per round)
        uint2 ra_n = make_uint2(0u, 0u); float4 rk_n = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < n_lt) { ra_n = rec0[lane]; rk_n = rowk[l0 + lane]; }
        for (int u = 0; u < n_regs; ++u) {
            const int e = u * 64 + lane;
            const bool in = e < n_lt;
            const uint2 ra = ra_n; const float4 rkc = rk_n;
            if (e + 64 < n_lt) { ra_n = rec0[e + 64]; rk_n = rowk[l0 + e + 64]; }
            bool active = false;
            if (in) {
                float lo, hi; bounds_rk(rkc, ra, lo, hi);
                active = ord_f32(hi) >= C;
            }
            if (in && !active && !compact) { rm_val[o + e] = -INFINITY; rm_arg[o + e] = 0; }
            // compact form: the active rows' (value, row | point << 16) side by side in row order (what S7 reads: a third of the rows), values also at their row (for the list's sums)
            const unsigned long long am = __ballot(active);
            const int my_slot = n_act + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0));
            n_act += (int)__popcll(am);
            uint32_t pts[4]; int cnt = 0; bool full = false;
            if (active) {
                const uint32_t dsc = ra.y;
                full = (dsc >> 20) & 1u;
                const uint32_t hp = (dsc >> 21) & 1u;                   // the primary half: {its groups} x {its slots}
                const uint32_t tt[2] = {dsc & 63u, (dsc >> 6) & 63u}, kk[2] = {(dsc >> 12) & 7u, (dsc >> 15) & 7u};
                const int nt = 1 + (int)((dsc >> 18) & 1u), nk = 1 + (int)((dsc >> 19) & 1u);
                auto add = [&](uint32_t grp, uint32_t slot, uint32_t hh, bool take) {
                    const uint32_t rr = slot + 8u * (grp & 1u);
                    const uint32_t p = 32u * (grp >> 1) + (rr & 3u) + 8u * (rr >> 2) + 4u * hh;
                    if (take && p < (uint32_t)n_rt) {
#pragma unroll
                        for (int z = 0; z < 4; ++z) if (z == cnt) pts[z] = p;
                        ++cnt;
                    }
                };
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) add(tt[a], kk[b], hp, a < nt && b < nk);
                add((dsc >> 23) & 63u, dsc >> 29, hp ^ 1u, (dsc >> 22) & 1u);   // the other half's best cell, when within reach
                full = full || cnt > 4;                                 // five candidate cells (a full primary half and the other half): every point instead
                if (full) cnt = 0;
            }
            st_active += (unsigned long long)__popcll(__ballot(active));
            // rows whose candidates the bound pass could not pin down (or forced rows): every point, exactly — the whole wave per row
            unsigned long long fm = __ballot(active && full);
            while (fm) {
                const int src = (int)__ffsll((long long)fm) - 1;
                fm &= fm - 1;
                const int row = u * 64 + src;
                // std::max_element (matcher.cpp:730): the first point's value stands until a STRICTLY greater one comes.  A NaN similarity never compares
                // greater, and a NaN at point 0 (a NaN in the latent row: every similarity of the row is NaN) is never beaten; a row of -inf
                // (an infinite or overflowing descriptor) keeps point 0 as well.
                // The row's descriptor goes through the wave's (idle: values only live inside flush()) value buffer: every trip of the loop used to fetch its 96 floats again — 24 vector loads of
                // ONE address per lane and trip (13 trips per row): the texture path, not the arithmetic, was what a full row cost.  From LDS they are 24 broadcast reads.
                if (lane < 24) reinterpret_cast<float4*>(W.val)[lane] = reinterpret_cast<const float4*>(des + (size_t)row * kDes)[lane];
                RF_WSYNC();
                float bv = -INFINITY, v_first = 0.0f; int bp = 0x7fffffff;
                for (int p = lane; p < n_rt; p += 64) { const float v = exact_sim_lds(p); if (p == lane) v_first = v; if (v > bv) { bv = v; bp = p; } }
                RF_WSYNC();
                rf_argmax(bv, bp);
                const float s0 = __shfl(v_first, 0);                    // n_rt >= 1: lane 0 evaluated point 0
                if (s0 != s0 || bp == 0x7fffffff) { bv = s0; bp = 0; }
                const int sl = __shfl(my_slot, src);
                if (lane == 0) { if (compact) { rm_cv[o + sl] = bv; rm_arg[o + sl] = row | (bp << 16); } else { rm_val[o + row] = bv; rm_arg[o + row] = bp; } }
                ++st_full;
            }
            // append this round's items, rows in ascending order (a row's items stay adjacent)
            int incl = cnt;                                             // inclusive prefix sum over the lanes: four row_shr steps inside the rows of 16 lanes, then the totals of the rows below (no LDS permutes)
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
            const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47), t3 = __builtin_amdgcn_readlane(incl, 63);
            incl += lane < 16 ? 0 : lane < 32 ? t0 : lane < 48 ? t0 + t1 : t0 + t1 + t2;
            const int total = t0 + t1 + t2 + t3;
            if (n_items + total > kRfItems) flush();
            const int base = n_items + incl - cnt;
#pragma unroll
            for (int z = 0; z < 4; ++z) if (z < cnt) { W.row[base + z] = (unsigned short)e; W.pt[base + z] = (unsigned short)pts[z]; W.slot[base + z] = (unsigned short)my_slot; }
            n_items += total;
        }
        flush();
        if (compact && lane == 0) rm_n[task] = n_act;
        if (stats && lane == 0) {
            atomicAdd(stats + 0, 1ull); atomicAdd(stats + 1, (unsigned long long)n_lt); atomicAdd(stats + 2, st_active);
            atomicAdd(stats + 3, st_items); atomicAdd(stats + 4, st_full);
        }
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
hipError_t launch_mf_codebook(const float* codewords, void* cw16, float* cwn, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mf_codebook, dim3(kM), dim3(kK), 0, stream, codewords, (uint4*)cw16, cwn);
    return hipGetLastError();
}

hipError_t launch_mf_tiles(const GalleryDev& g, const int32_t* q_blk, const float* cwn, void* codes_p, float* nrm_p, void* tile_meta, hipStream_t stream)
{
    if (g.G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mf_tiles, dim3(g.G), dim3(64), 0, stream, g, q_blk, cwn, (uint4*)codes_p, nrm_p, (int2*)tile_meta);
    return hipGetLastError();
}

hipError_t launch_mf_rows(const float* lt_des, int n_rows, int n_rb, const float* codewords, const float* cwn, void* bfrag, void* rowk, hipStream_t stream)
{
    if (n_rb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mf_rows, dim3(n_rb * 32), dim3(256), 0, stream, lt_des, n_rows, codewords, cwn, (_Float16*)bfrag, (float4*)rowk);
    return hipGetLastError();
}

hipError_t launch_tex_refine(const QueryDev& q, const GalleryDev& g, const float* codewords, const void* rec, const void* rowk, int R_pad, int all_rows,
                             float* rm_val, int32_t* rm_arg, unsigned long long* stats, float* rm_cv, int32_t* rm_n, hipStream_t stream)
{
    const long long n_tasks = (long long)q.nq * g.G;
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7ffffff0LL || !g.task_ctr) return hipErrorInvalidValue;
    hipError_t e0 = hipMemsetAsync(g.task_ctr + 2, 0, 4, stream);
    if (e0 != hipSuccess) return e0;
    const int grid = (int)std::min<long long>(256, (n_tasks + kRfWaves - 1) / kRfWaves);
    hipLaunchKernelGGL(k_tex_refine, dim3(grid), dim3(kRfWaves * 64), 0, stream, q, g, codewords, (const uint2*)rec, (const float4*)rowk, R_pad, all_rows,
                       rm_val, rm_arg, g.task_ctr + 2, stats, rm_cv, rm_n);
    return hipGetLastError();
}

}  // namespace afis
