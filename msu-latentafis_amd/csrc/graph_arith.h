// graph_arith.h — the arithmetic of the graph kernels' packed paths (graph.hip), in a header of its own so that
// tools/ubench/graph_arith.hip checks exactly this code against the straightforward evaluation (exhaustively where the domain allows).
#pragma once
#include "afis_device.h"

namespace afis {

typedef short ga_v2s16 __attribute__((ext_vector_type(2)));
typedef unsigned short ga_v2u16 __attribute__((ext_vector_type(2)));

typedef _Float16 ga_v2h __attribute__((ext_vector_type(2)));

// The packed paths keep a point as two half2 words (x, y as fp16: every integer of [0, 2047] is an fp16, and so is every difference of two).
__device__ __forceinline__ int pack_h2(int x, int y) { const ga_v2h v = {(_Float16)(float)x, (_Float16)(float)y}; return __builtin_bit_cast(int, v); }
__device__ __forceinline__ void unpack_h2(int w, int& x, int& y) { const ga_v2h v = __builtin_bit_cast(ga_v2h, w); x = (int)(float)v.x; y = (int)(float)v.y; }

// n = dx^2 + dy^2 of the latent (n1) and of the rolled (n2) point pair from the packed words, as floats: one v_pk_add_f16 (exact: the
// differences are integers of [-2047, 2047]) and one v_dot2_f32_f16 per side (fp32 products and sum, exact below 2^24) — the fp32 value
// arrives without the two v_cvt_f32_i32 the 16-bit integer form (v_pk_sub_i16 + v_dot2_i32_i16) needed.  Checked on the device against
// the integer evaluation for every (dx, dy) of [-2047, 2047]^2 (afis_debug_graph_arith, part 3).
__device__ __forceinline__ float diff_n(int a, int o)
{
    const ga_v2h d = __builtin_bit_cast(ga_v2h, a) - __builtin_bit_cast(ga_v2h, o);
    return __builtin_amdgcn_fdot2(d, d, 0.0f, false);
}
// Both sides of a pair.  The three-operand form with an inline 0 accumulator: the builtin selects the accumulate-in-place v_dot2c and pays a
// v_mov 0 per call.  A DOT result needs wait states before a VALU reads it, which the compiler cannot know about for an asm's outputs: s_nop 2.
__device__ __forceinline__ void pair_n(int2 a, int2 o, float& n1, float& n2)
{
    const ga_v2h dl = __builtin_bit_cast(ga_v2h, a.x) - __builtin_bit_cast(ga_v2h, o.x);
    const ga_v2h dr = __builtin_bit_cast(ga_v2h, a.y) - __builtin_bit_cast(ga_v2h, o.y);
    asm("v_dot2_f32_f16 %0, %2, %2, 0\n\tv_dot2_f32_f16 %1, %3, %3, 0\n\ts_nop 2" : "=&v"(n1), "=v"(n2) : "v"(dl), "v"(dr));
}
// |d| < 50 for all four coordinate differences (matcher.cpp:1257); only lists with a block coordinate outside [0, 49] ask (never a real template)
__device__ __forceinline__ bool tex_in_range(int2 a, int2 o)
{
    const ga_v2h dl = __builtin_bit_cast(ga_v2h, a.x) - __builtin_bit_cast(ga_v2h, o.x);
    const ga_v2h dr = __builtin_bit_cast(ga_v2h, a.y) - __builtin_bit_cast(ga_v2h, o.y);
    const float m = fmaxf(fmaxf(fabsf((float)dl.x), fabsf((float)dl.y)), fmaxf(fabsf((float)dr.x), fabsf((float)dr.y)));
    return m < 50.0f;
}

// RN(sqrt(x)) for an INTEGER-valued x in [0, 2 * 2047^2] (the packed paths' dx^2 + dy^2): v_rsq_f32 (1 ulp) and one fma
// correction step.  Equal to sqrt_rn_pos for every such integer (exhaustive check).  x = 0: rsq = inf, clamped to 1 (rsq <= 1 for
// x >= 1 anyway), and the chain gives 0.  One transcendental + 4 instructions; sqrt_rn_pos: one + 9.
__device__ __forceinline__ float sqrt_rn_int(float x)
{
    // min(rsq, 1) written as clamp(rsq, 0, 1) (rsq >= 0): the compiler folds it into the instruction's own clamp bit — one issue slot less than v_min_f32.
    // (Not an asm: x comes out of a DOT instruction, whose wait states before a VALU read the compiler only inserts for instructions it can see.)
    const float y = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_rsqf(x), 0.0f), 1.0f);
    const float r0 = x * y;
    const float e = __builtin_fmaf(-r0, r0, x);
    return __builtin_fmaf(e, 0.5f * y, r0);
}

// "H != 0" on the packed paths without a square root.  With a = sqrt n1, b = sqrt n2, s = n1 + n2 and thr = 30 px (minutiae) or
// 30/16 blocks (texture, whose table entry is 16 RN(sqrt n) exactly: scaling by a power of two commutes with every rounding):
//   |a - b| < thr  <=>  u < 0  or  u^2 < n1 n2,   u = (s - thr^2) / 2       (u is exact in fp32; u |u| folds the sign test in).
// The reference compares the ROUNDED |RN a - RN b| with thr; that differs from the real |a - b| by at most 1.5 ulp(max(a, b)), and
// t = u^2 - n1 n2 = (|a-b|^2 - thr^2)((a+b)^2 - thr^2)/4, so the two decisions can only differ where |t| <= 3 * 2^-23 s^2 (uses
// (a+b)^3 <= (a+b)^4 / thr <= 4 s^2 / thr on u >= 0); the fp32 evaluation of t adds at most 2^-23 s^2.  The band used is
// 2^-18 u^2 + 2^-20 thr^4 >= 2^-21 s^2 (s = 2u + thr^2), a function of u alone; outside it the sign of the computed t decides, inside
// it (about 1e-4 of the pairs) the rounded roots are compared.  Checked against the rounded-root predicate on every (n1, n2) in
// [0, 4802]^2 and on 1.3e10 near-threshold / random minutiae pairs: no disagreement outside the band.  8 instructions; the
// rounded-root test with its own slack: 5 + two transcendentals (4 issue slots each).
template <bool TEX>
__device__ __forceinline__ int pair_compatible_alg(float n1, float n2)   // 1 / 0: decided; 2: inside the band
{
    constexpr float c = TEX ? 3.515625f : 900.0f;                          // thr^2
    const float s = n1 + n2;
    const float u = __builtin_fmaf(s, 0.5f, -0.5f * c);
    const float uu = u * fabsf(u);
    const float p = n1 * n2;
    const float t = uu - p;
    const float band = __builtin_fmaf(fabsf(uu), 3.814697265625e-6f, c * c * 9.5367431640625e-7f);
    if (!(fabsf(t) > band)) return 2;
    return t < 0.0f ? 1 : 0;
}
// the same test split in two for callers that evaluate several independent pairs at once: pair_compatible_t gives the sign test and
// says whether the pair is inside the band; pair_compatible_exact is the rounded-root comparison for those that are
template <bool TEX>
__device__ __forceinline__ bool pair_compatible_t(float n1, float n2, bool& near)
{
    constexpr float c = TEX ? 3.515625f : 900.0f;
    const float s = n1 + n2;
    const float u = __builtin_fmaf(s, 0.5f, -0.5f * c);
    const float uu = u * fabsf(u);
    const float p = n1 * n2;
    const float t = uu - p;
    const float band = __builtin_fmaf(fabsf(uu), 3.814697265625e-6f, c * c * 9.5367431640625e-7f);
    near = !(fabsf(t) > band);
    return t < 0.0f;
}
template <bool TEX>
__device__ __forceinline__ bool pair_compatible_exact(float n1, float n2)
{
    const float d = fabsf(sqrt_rn_pos(n1) - sqrt_rn_pos(n2));
    return TEX ? 16.0f * d < 30.0f : d < 30.0f;
}
template <bool TEX>
__device__ __forceinline__ bool pair_compatible_n(float n1, float n2)
{
    constexpr float c = TEX ? 3.515625f : 900.0f;
    const float s = n1 + n2;
    const float u = __builtin_fmaf(s, 0.5f, -0.5f * c);
    const float uu = u * fabsf(u);
    const float p = n1 * n2;
    const float t = uu - p;
    const float band = __builtin_fmaf(fabsf(uu), 3.814697265625e-6f, c * c * 9.5367431640625e-7f);
    bool hit = t < 0.0f;
    const bool near = !(fabsf(t) > band);
    // one UNIFORM branch around the rare case (a lane of the wave inside the band: once in ~150 calls) instead of a divergent if / else
    // per call, whose exec bookkeeping costs as many scalar instructions as the test itself costs vector ones
    if (__builtin_amdgcn_ballot_w64(near) != 0ull) {
        if (near) {
            const float d = fabsf(sqrt_rn_pos(n1) - sqrt_rn_pos(n2));
            hit = TEX ? 16.0f * d < 30.0f : d < 30.0f;
        }
    }
    return hit;
}

}  // namespace afis
