// Which ingredient of k_adc_mfma's tile step keeps vector work from hiding under matrix work?  Two waves per SIMD, one stream per wave:
// 12 v_mfma_f32_32x32x16_f16 (two accumulator chains) with 4 VALU after each MFMA (48 per unit), built up ingredient by ingredient:
//   0  MFMA operands constant, VALU on private registers                               (the reference point: 800 cycles in mfma_valu_overlap.hip)
//   1  + MFMA A operands from a rotating set of 6 register quads, B from 12
//   2  + the VALU read the OTHER accumulator set (v_max3 / v_med3 chains shaped like the tracking), MFMAs alternate between two sets per unit
//   3  + 10 ds_read_b128 per unit into the A operand registers of the next unit (conflict-free addresses that change every unit)
//   4  + first MFMA of each chain takes its C operand from a third register block (the point terms)
//   5  + every 4 units a decode phase (4 ds_read_b128 of random slots, 3 ds_write_b128) and a workgroup barrier
//   6  + every 26 units a "template end": ~80 VALU, one 8-byte store per lane, state reset (a uniform branch in the stream)
//   7  as 6, but the decode's LDS operations are spread between the MFMAs of the 4 units and the barrier sits in the middle of a unit's MFMA stream
//      (three-slot ring: the data decoded in stage s is read from stage s + 2 on, so any one barrier per stage orders everything)
//   8  no LDS operands at all: every wave loads the tile's operand groups (6 KB per tile, the same addresses for the 8 waves of a workgroup, a
//      fresh tile every unit: L1 / L2 traffic) straight from global memory one unit ahead; point terms through a wave-private LDS slice; no decode,
//      no workgroup barrier; template-end branch as in 6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
struct Trk { float m[2][8], tb[2], ts[2], tu[2]; };
__device__ __forceinline__ void track_part(Trk& t, int blk, const floatx16& X, unsigned gid, int part)   // the tracking cut in 6 parts of 4 VALU
{
    if (part == 0) { t.m[blk][0] = max3f(t.m[blk][0], X[0], X[8]); t.m[blk][1] = max3f(t.m[blk][1], X[1], X[9]); t.m[blk][2] = max3f(t.m[blk][2], X[2], X[10]); t.m[blk][3] = max3f(t.m[blk][3], X[3], X[11]); }
    if (part == 1) { t.m[blk][4] = max3f(t.m[blk][4], X[4], X[12]); t.m[blk][5] = max3f(t.m[blk][5], X[5], X[13]); t.m[blk][6] = max3f(t.m[blk][6], X[6], X[14]); t.m[blk][7] = max3f(t.m[blk][7], X[7], X[15]); }
    if (part == 2) { float lo = max3f(X[0], X[1], X[2]); lo = max3f(lo, X[3], X[4]); lo = max3f(lo, X[5], X[6]); t.tu[blk] = fmaxf(lo, X[7]); }
    if (part == 3) { float hi = max3f(X[8], X[9], X[10]); hi = max3f(hi, X[11], X[12]); hi = max3f(hi, X[13], X[14]); t.ts[blk] = fmaxf(hi, X[15]); }
    if (part == 4) { const float el = __uint_as_float((__float_as_uint(t.tu[blk]) & ~63u) | gid); t.tb[blk] = __builtin_amdgcn_fmed3f(t.tb[blk], el, t.m[blk][0]); t.m[blk][1] = fmaxf(t.m[blk][1], el); }
    if (part == 5) { const float eh = __uint_as_float((__float_as_uint(t.ts[blk]) & ~63u) | gid); t.tb[blk] = __builtin_amdgcn_fmed3f(t.tb[blk], eh, t.m[blk][2]); t.m[blk][3] = fmaxf(t.m[blk][3], eh); }
}
template <int LVL>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, const uint4* __restrict__ gsrc)
{
    __shared__ uint4 s_l[16][64];
    __shared__ uint4 s_cw[4096];
    __shared__ uint4 s_ring[2][48][32];
    __shared__ float s_priv[8][2][32];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16 * 64; i += 512) (&s_l[0][0])[i] = make_uint4(0x3c003c00u, 0x38003800u + i, 0x34003400u, 0x30003000u);
    for (int i = tid; i < 4096; i += 512) s_cw[i] = make_uint4(i, i * 3, i * 5, 0);
    unsigned code = tid * 2654435761u;
    uint4 dw[4] = {};
    half8 bq[12], aq[2][6];
    for (int i = 0; i < 12; ++i) for (int e = 0; e < 8; ++e) bq[i][e] = (_Float16)(0.01f * (lane + i + e));
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) aq[s][i][e] = (_Float16)(0.02f * (lane + i + e + s));
    floatx16 acc[2][2], nrm;
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = acc[0][1][r] = acc[1][0][r] = acc[1][1][r] = 0.f; nrm[r] = -1.f - r; }
    Trk t; for (int b = 0; b < 2; ++b) { for (int q = 0; q < 8; ++q) t.m[b][q] = -1e30f; t.tb[b] = t.ts[b] = t.tu[b] = -1e30f; }
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = tid + i;
    const float x = out[0], y = out[1];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {                          // two units per iteration: static set indices
            const unsigned gid = (unsigned)((2 * it + par) & 31) * 2u;
            if (LVL == 8) {
                const uint4* tsrc = gsrc + ((size_t)blockIdx.x * 4096 + (size_t)((2 * it + par) & 4095)) * 448;     // this workgroup's tile stream: 7 x 64 x 16 B per tile
#pragma unroll
                for (int q = 0; q < 6; ++q) aq[par ^ 1][q] = __builtin_bit_cast(half8, tsrc[q * 64 + lane]);
                if (lane < 32) s_priv[tid >> 6][par][lane] = reinterpret_cast<const float*>(tsrc + 6 * 64)[lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 w = *reinterpret_cast<const float4*>(&s_priv[tid >> 6][par ^ 1][8 * q + 4 * (lane >> 5)]); nrm[4 * q] = w.x; nrm[4 * q + 1] = w.y; nrm[4 * q + 2] = w.z; nrm[4 * q + 3] = w.w; }
            } else if (LVL >= 3) {
#pragma unroll
                for (int q = 0; q < 6; ++q) aq[par ^ 1][q] = __builtin_bit_cast(half8, s_l[q][(lane + it) & 63]);
                if (LVL >= 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const uint4 w = s_l[6 + q][(lane + it + par) & 63]; nrm[4 * q] = __uint_as_float(w.x); nrm[4 * q + 1] = __uint_as_float(w.y); nrm[4 * q + 2] = __uint_as_float(w.z); nrm[4 * q + 3] = __uint_as_float(w.w); }
                }
            }
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const half8 a = LVL >= 1 ? aq[par][kk] : aq[0][0];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const half8 b = LVL >= 1 ? bq[blk * 6 + kk] : bq[0];
                    floatx16& D = LVL >= 2 ? acc[par][blk] : acc[0][blk];
                    D = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, (LVL >= 4 && kk == 0) ? nrm : D, 0, 0, 0);
                    if (LVL >= 2) track_part(t, blk, acc[par ^ 1][blk], gid, kk);
                    if (LVL == 7) {
                        const int unit = (it & 1) * 2 + par;             // unit of the stage (0..3)
                        uint4 (&dst)[48][32] = s_ring[(it >> 1) & 1];
                        if (blk == 0 && kk == 1 && unit < 2) { dw[2 * unit] = s_cw[((2 * unit) * 1024 + ((code >> (16 * unit)) & 255u) * 4 + (tid & 3)) & 4095]; dw[2 * unit + 1] = s_cw[((2 * unit + 1) * 1024 + ((code >> (16 * unit + 8)) & 255u) * 4 + (tid & 3)) & 4095]; }
                        if (blk == 0 && kk == 3 && unit == 0) __syncthreads();
                        if (blk == 1 && kk == 2 && unit == 1) dst[(tid >> 5) * 3 + 0][tid & 31] = make_uint4(dw[0].x, dw[0].y, dw[0].z, dw[1].x);
                        if (blk == 1 && kk == 2 && unit == 2) dst[(tid >> 5) * 3 + 1][tid & 31] = make_uint4(dw[1].y, dw[1].z, dw[2].x, dw[2].y);
                        if (blk == 1 && kk == 2 && unit == 3) { dst[(tid >> 5) * 3 + 2][tid & 31] = make_uint4(dw[2].z, dw[3].x, dw[3].y, dw[3].z); code = code * 1664525u + 1013904223u; }
                    }
                    else {
#pragma unroll
                        for (int z = 0; z < 4; ++z) { const int i = (kk * 8 + blk * 4 + z) & 7; v[i] = (z & 1) ? __builtin_amdgcn_fmed3f(v[i], x, y) : max3f(v[i], x, y); }
                    }
                }
            }
            if (LVL == 6 || LVL == 7 || LVL == 8) {
                if (((2 * it + par) % 26) == 25) {                  // uniform
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        float b3 = -1e30f, s3 = -1e30f, u3 = -1e30f;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const float e = __uint_as_float((__float_as_uint(t.m[blk][q]) & ~7u) | (unsigned)q); u3 = __builtin_amdgcn_fmed3f(s3, e, u3); s3 = __builtin_amdgcn_fmed3f(b3, s3, e); b3 = fmaxf(b3, e); }
                        const float thr = fminf(t.tb[blk], b3) - 0.01f;
                        const unsigned desc = (__float_as_uint(t.tb[blk]) & 63u) | ((__float_as_uint(b3) & 7u) << 12) | ((s3 >= thr ? 1u : 0u) << 19) | ((u3 >= thr ? 1u : 0u) << 20);
                        reinterpret_cast<uint2*>(out + 2 + 256 * 512)[((size_t)blockIdx.x * 2 + blk) * 512 + tid] = make_uint2(__float_as_uint(b3), desc);
#pragma unroll
                        for (int q = 0; q < 8; ++q) t.m[blk][q] = -1e30f;
                        t.tb[blk] = t.ts[blk] = t.tu[blk] = -1e30f;
                    }
                }
            }
        }
        if ((LVL == 5 || LVL == 6) && (it & 1)) {                       // every 4 units
            code = code * 1664525u + 1013904223u;
            uint4 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = s_cw[(i * 1024 + ((code >> (8 * i)) & 255u) * 4 + (tid & 3)) & 4095];
            uint4 (&dst)[48][32] = s_ring[(it >> 1) & 1];
            dst[(tid >> 5) * 3 + 0][tid & 31] = make_uint4(w[0].x, w[0].y, w[0].z, w[1].x);
            dst[(tid >> 5) * 3 + 1][tid & 31] = make_uint4(w[1].y, w[1].z, w[2].x, w[2].y);
            dst[(tid >> 5) * 3 + 2][tid & 31] = make_uint4(w[2].z, w[3].x, w[3].y, w[3].z);
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sacc = 0;
    for (int s = 0; s < 2; ++s) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sacc += acc[s][b][r];
    for (int b = 0; b < 2; ++b) { for (int q = 0; q < 8; ++q) sacc += t.m[b][q]; sacc += t.tb[b] + t.ts[b] + t.tu[b]; }
    for (int i = 0; i < 8; ++i) sacc += v[i];
    out[2 + blockIdx.x * blockDim.x + tid] = sacc;
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}
template <int LVL> double run()
{
    float* d; unsigned long long* dc;
    (void)hipMalloc(&d, (2 + 256 * 512) * 4 + 256 * 2 * 512 * 8); (void)hipMalloc(&dc, 256 * 8 * 8); (void)hipMemset(d, 0, 8);
    const int iters = 1500;
    static uint4* gsrc = nullptr;
    if (!gsrc) { (void)hipMalloc(&gsrc, (size_t)256 * 4096 * 448 * 16); (void)hipMemset(gsrc, 0x3c, (size_t)256 * 4096 * 448 * 16); }
    hipLaunchKernelGGL((k<LVL>), dim3(256), dim3(512), 0, 0, d, dc, 30, gsrc);
    hipLaunchKernelGGL((k<LVL>), dim3(256), dim3(512), 0, 0, d, dc, iters, gsrc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    (void)hipFree(d); (void)hipFree(dc);
    return (double)h[h.size() / 2] / (iters * 2);
}
int main()
{
    printf("{\"benchmark\": \"tools/ubench/mfma_stream_model.hip\", \"unit\": \"12 v_mfma_f32_32x32x16_f16 + 48 VALU per wave, two waves per SIMD; 768 cycles = the matrix pipe's share\", \"cycles_per_unit_per_wave\": {"
           "\"0_constant_operands_private_valu\": %.0f, \"1_rotating_operands\": %.0f, \"2_valu_tracks_other_accumulator_set\": %.0f, \"3_operands_from_lds_one_unit_ahead\": %.0f, \"4_c_operand_from_lds\": %.0f, \"5_decode_phase_and_barrier_every_4_units\": %.0f, \"6_template_end_branch\": %.0f, \"7_decode_between_the_mfmas_barrier_mid_stream\": %.0f, \"8_operands_from_global_memory_no_lds_no_barrier\": %.0f}}\n",
           run<0>(), run<1>(), run<2>(), run<3>(), run<4>(), run<5>(), run<6>(), run<7>(), run<8>());
    return 0;
}
