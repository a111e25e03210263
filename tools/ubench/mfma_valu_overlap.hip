// Does vector work hide under matrix work on one SIMD of MI355X?  One workgroup of 8 waves per CU (two per SIMD: waves w and w + 4 share a SIMD).
//   mode 0: every wave: MFMA bursts only          mode 1: every wave: VALU bursts only (v_max3_f32 / v_med3_f32 on registers)
//   mode 2: waves 0-3 MFMA only, waves 4-7 VALU only (perfectly complementary roles)
//   mode 3: every wave alternates a burst of 12 MFMAs and a burst of nv VALU; the two waves of a SIMD half a period apart
//   mode 4: every wave: nv / 12 VALU after each MFMA (interleaved stream)
// Reports shader cycles (s_memtime, median over waves) per (12 MFMAs + nv VALU) unit of ONE wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define MFMA(c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
template <int NV> __device__ __forceinline__ void valu_burst(float (&v)[8], float x, float y)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float r;
        if (i & 1) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v[i & 7]), "v"(x), "v"(y));
        else asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v[i & 7]), "v"(x), "v"(y));
        v[i & 7] = r;
    }
}
template <int NV> __device__ __forceinline__ void valu_burst_acc(float (&v)[8], const floatx16& c0, const floatx16& c1)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float r; const float x = (i & 16) ? c1[i & 15] : c0[i & 15], y = (i & 16) ? c0[(i + 5) & 15] : c1[(i + 5) & 15];
        if (i & 1) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v[i & 7]), "v"(x), "v"(y));
        else asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v[i & 7]), "v"(x), "v"(y));
        v[i & 7] = r;
    }
}
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ uint4 s_l[11][64];
    for (int i = threadIdx.x; i < 11 * 64; i += 512) (&s_l[0][0])[i] = make_uint4(i, i + 1, i + 2, i + 3);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    half8 a, b; floatx16 c0, c1;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 1; }
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float x = out[0], y = out[1];
    const bool late = wave >= 4;
    uint4 ldA[6], ldB[6], nv5[5];
    for (int q = 0; q < 6; ++q) { ldA[q] = make_uint4(0x3c003c00u, 0x38003800u, threadIdx.x, q); ldB[q] = ldA[q]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 2 && !late)) { for (int k2 = 0; k2 < 6; ++k2) { MFMA(c0); MFMA(c1); } }
        else if (MODE == 1 || (MODE == 2 && late)) valu_burst<NV>(v, x, y);
        else if (MODE == 3) {
            if (late) valu_burst<NV>(v, x, y);
            for (int k2 = 0; k2 < 6; ++k2) { MFMA(c0); MFMA(c1); }
            if (!late) valu_burst<NV>(v, x, y);
        } else if (MODE == 5) {
            if (late) valu_burst_acc<NV>(v, c0, c1);
            for (int k2 = 0; k2 < 6; ++k2) { MFMA(c0); MFMA(c1); }
            if (!late) valu_burst_acc<NV>(v, c0, c1);
        } else if (MODE == 6) {
            if (late) valu_burst<NV>(v, x, y);
            for (int k2 = 0; k2 < 6; ++k2) { MFMA(c0); MFMA(c1); }
            uint4 ld[11];
#pragma unroll
            for (int q = 0; q < 11; ++q) ld[q] = s_l[q][(threadIdx.x + it) & 63];
#pragma unroll
            for (int q = 0; q < 11; ++q) v[q & 7] += __uint_as_float(ld[q].x ^ ld[q].w);
            if (!late) valu_burst<NV>(v, x, y);
        } else if (MODE == 7) {
            if (late) valu_burst_acc<NV>(v, c0, c1);
            for (int k2 = 0; k2 < 6; ++k2) { MFMA(c0); MFMA(c1); }
            uint4 ld[11];
#pragma unroll
            for (int q = 0; q < 11; ++q) ld[q] = s_l[q][(threadIdx.x + it) & 63];
#pragma unroll
            for (int q = 0; q < 11; ++q) v[q & 7] += __uint_as_float(ld[q].x ^ ld[q].w);
            if (!late) valu_burst_acc<NV>(v, c0, c1);
        } else if (MODE == 8 || MODE == 9) {
            // operands of the unit from LDS, read one unit ahead into ping-pong registers; MODE 8: VALU between the MFMAs, MODE 9: VALU burst after them
            uint4 (&cur)[6] = (it & 1) ? ldB : ldA; uint4 (&nxt)[6] = (it & 1) ? ldA : ldB;
#pragma unroll
            for (int q = 0; q < 6; ++q) nxt[q] = s_l[q][(threadIdx.x + it) & 63];
#pragma unroll
            for (int q = 0; q < 5; ++q) nv5[q] = s_l[6 + q][(threadIdx.x + 2 * it) & 63];
#pragma unroll
            for (int k2 = 0; k2 < 6; ++k2) {
                const half8 aa = __builtin_bit_cast(half8, cur[k2]);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b, c0, 0, 0, 0); if (MODE == 8) valu_burst<NV / 12>(v, x, y);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b, c1, 0, 0, 0); if (MODE == 8) valu_burst<NV / 12>(v, x, y);
            }
            if (MODE == 9) valu_burst<NV>(v, x, y);
#pragma unroll
            for (int q = 0; q < 5; ++q) v[q] += __uint_as_float(nv5[q].x);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 6; ++k2) { MFMA(c0); valu_burst<NV / 12>(v, x, y); MFMA(c1); valu_burst<NV / 12>(v, x, y); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sacc = 0; for (int r = 0; r < 16; ++r) sacc += c0[r] + c1[r];
    for (int i = 0; i < 8; ++i) sacc += v[i];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = sacc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int NV> double run()
{
    float* d; unsigned long long* dc;
    (void)hipMalloc(&d, (2 + 256 * 512) * 4); (void)hipMalloc(&dc, 256 * 8 * 8);
    (void)hipMemset(d, 0, 8);
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(512), 0, 0, d, dc, 50);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(512), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    (void)hipFree(d); (void)hipFree(dc);
    return (double)h[h.size() / 2] / iters;
}
template <int NV> void row(bool last)
{
    printf("  {\"valu_per_12_mfma\": %d, \"cycles_per_unit\": {\"mfma_only\": %.0f, \"valu_only\": %.0f, \"mfma_waves_beside_valu_waves\": %.0f, \"bursts_half_a_period_apart\": %.0f, \"interleaved_stream\": %.0f, \"half_period_valu_reads_accumulators\": %.0f, \"half_period_plus_11_lds_reads\": %.0f, \"half_period_acc_and_lds\": %.0f, \"interleaved_with_operands_from_lds_one_unit_ahead\": %.0f, \"burst_after_mfmas_with_operands_from_lds\": %.0f}}%s\n",
           NV, run<0, NV>(), run<1, NV>(), run<2, NV>(), run<3, NV>(), run<4, NV>(), run<5, NV>(), run<6, NV>(), run<7, NV>(), run<8, NV>(), run<9, NV>(), last ? "" : ",");
}
int main()
{
    printf("{\"benchmark\": \"tools/ubench/mfma_valu_overlap.hip\", \"what\": \"shader cycles one wave needs for a unit of 12 v_mfma_f32_32x32x16_f16 (two accumulators) + N v_max3/v_med3, two waves per SIMD, five arrangements; "
           "a unit of 12 MFMAs alone is 384 matrix-pipe cycles, shared by the two waves of the SIMD\", \"rows\": [\n");
    row<24>(false); row<48>(false); row<96>(true);
    printf("]}\n");
    return 0;
}
