// Model of k_adc_mfma's tile step on one CU, to find what keeps the matrix pipe at half speed: per unit and wave 11 ds_read_b128 (the tile's operands and
// point terms), 12 v_mfma_f32_32x32x16_f16 in two accumulator chains fed by those reads, 48 v_max3/v_med3 that READ the accumulators (real data
// dependence, the tracking's chain shapes).  W waves per workgroup (8 = two per SIMD, 12 = three).  Arrangements:
//   A plain: reads, MFMAs, tracking (what the compiler makes of the straightforward loop)
//   B reads of the NEXT unit issued before this unit's tracking (one unit ahead, ping-pong registers)
//   C as B, late half of the waves tracks the previous unit before its MFMAs (bursts half a period apart)
//   D as A without the LDS reads (operands constant)      E as A without the tracking     F as A without MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
struct Trk { float m[2][8], tb[2], ts[2], tu[2]; };
__device__ __forceinline__ void track(Trk& t, int blk, const floatx16& X, unsigned gid)
{
    float lo = max3f(X[0], X[1], X[2]), hi = max3f(X[8], X[9], X[10]);
    lo = max3f(lo, X[3], X[4]); hi = max3f(hi, X[11], X[12]);
    lo = max3f(lo, X[5], X[6]); hi = max3f(hi, X[13], X[14]);
    lo = fmaxf(lo, X[7]); hi = fmaxf(hi, X[15]);
    const float el = __uint_as_float((__float_as_uint(lo) & ~63u) | gid), eh = __uint_as_float((__float_as_uint(hi) & ~63u) | (gid + 1u));
    t.tu[blk] = __builtin_amdgcn_fmed3f(t.ts[blk], el, t.tu[blk]); t.ts[blk] = __builtin_amdgcn_fmed3f(t.tb[blk], t.ts[blk], el); t.tb[blk] = fmaxf(t.tb[blk], el);
    t.tu[blk] = __builtin_amdgcn_fmed3f(t.ts[blk], eh, t.tu[blk]); t.ts[blk] = __builtin_amdgcn_fmed3f(t.tb[blk], t.ts[blk], eh); t.tb[blk] = fmaxf(t.tb[blk], eh);
#pragma unroll
    for (int k = 0; k < 8; ++k) t.m[blk][k] = max3f(t.m[blk][k], X[k], X[k + 8]);
}
template <int W, int ARR>
__global__ __launch_bounds__(W * 64) void k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ uint4 s_a[4][12][32];
    __shared__ float s_n[4][32];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 12 * 32; i += W * 64) (&s_a[0][0][0])[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x34003400u, 0x30003000u);
    if (tid < 128) (&s_n[0][0])[tid] = -1.0f - tid * 0.001f;
    half8 bf[2][6];
    for (int b = 0; b < 2; ++b) for (int kk = 0; kk < 6; ++kk) for (int e = 0; e < 8; ++e) bf[b][kk][e] = (_Float16)(0.01f * (lane + kk + e + b));
    Trk t;
    for (int b = 0; b < 2; ++b) { for (int k2 = 0; k2 < 8; ++k2) t.m[b][k2] = -1e30f; t.tb[b] = t.ts[b] = t.tu[b] = -1e30f; }
    floatx16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = -1e30f;
    half8 afr[2][6]; floatx16 nrr[2];
    auto load = [&](int par, int j) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) { const float4 v = *reinterpret_cast<const float4*>(&s_n[j][8 * q4 + 4 * h]); nrr[par][4 * q4] = v.x; nrr[par][4 * q4 + 1] = v.y; nrr[par][4 * q4 + 2] = v.z; nrr[par][4 * q4 + 3] = v.w; }
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) afr[par][kk] = __builtin_bit_cast(half8, s_a[j][2 * kk + h][col]);
    };
    auto mfma = [&](int par) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[par][0], bf[0][0], nrr[par], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[par][0], bf[1][0], nrr[par], 0, 0, 0);
#pragma unroll
        for (int kk = 1; kk < 6; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[par][kk], bf[0][kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[par][kk], bf[1][kk], acc[1], 0, 0, 0);
        }
    };
    const bool late = (wave & (W == 8 ? 4 : 1)) != 0;
    __syncthreads();
    if (ARR == 1 || ARR == 2) load(0, 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int par = j & 1;
            const unsigned gid = (unsigned)((it * 4 + j) & 31) * 2u;
            if (ARR == 0) { load(0, j); mfma(0); track(t, 0, acc[0], gid); track(t, 1, acc[1], gid); }
            else if (ARR == 1) { mfma(par); load(par ^ 1, (j + 1) & 3); track(t, 0, acc[0], gid); track(t, 1, acc[1], gid); }
            else if (ARR == 2) {
                if (late) { track(t, 0, acc[0], gid); track(t, 1, acc[1], gid); }
                mfma(par); load(par ^ 1, (j + 1) & 3);
                if (!late) { track(t, 0, acc[0], gid); track(t, 1, acc[1], gid); }
            }
            else if (ARR == 3) { if (it == 0 && j == 0) load(0, 0); mfma(0); track(t, 0, acc[0], gid); track(t, 1, acc[1], gid); }
            else if (ARR == 4) { load(0, j); mfma(0); t.m[0][0] = max3f(t.m[0][0], acc[0][0], acc[0][15]); t.m[1][0] = max3f(t.m[1][0], acc[1][3], acc[1][12]); }
            else { load(0, j); acc[0] = nrr[0]; acc[1] = nrr[0]; acc[0][1] += (float)afr[0][0][0]; acc[1][2] += (float)afr[0][5][1]; track(t, 0, acc[0], gid); track(t, 1, acc[1], gid); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sacc = 0;
    for (int b = 0; b < 2; ++b) { for (int k2 = 0; k2 < 8; ++k2) sacc += t.m[b][k2]; sacc += t.tb[b] + t.ts[b] + t.tu[b]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
    if (lane == 0) cyc[blockIdx.x * W + wave] = t1 - t0;
}
template <int W, int ARR> double run()
{
    float* d; unsigned long long* dc;
    (void)hipMalloc(&d, 256 * W * 64 * 4); (void)hipMalloc(&dc, 256 * W * 8);
    const int iters = 1500;
    hipLaunchKernelGGL((k<W, ARR>), dim3(256), dim3(W * 64), 0, 0, d, dc, 30);
    hipLaunchKernelGGL((k<W, ARR>), dim3(256), dim3(W * 64), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> hh(256 * W);
    (void)hipMemcpy(hh.data(), dc, hh.size() * 8, hipMemcpyDeviceToHost);
    std::sort(hh.begin(), hh.end());
    (void)hipFree(d); (void)hipFree(dc);
    return (double)hh[hh.size() / 2] / (iters * 4);
}
template <int W> void row(bool last)
{
    printf("  {\"waves_per_workgroup\": %d, \"matrix_pipe_cycles_per_unit_for_all_waves_of_a_simd\": %d, \"cycles_per_unit_per_wave\": {\"A_plain\": %.0f, \"B_reads_one_unit_ahead\": %.0f, \"C_B_plus_half_period_offset\": %.0f, "
           "\"D_no_lds_reads\": %.0f, \"E_no_tracking\": %.0f, \"F_no_mfma\": %.0f}}%s\n", W, 384 * W / 4, run<W, 0>(), run<W, 1>(), run<W, 2>(), run<W, 3>(), run<W, 4>(), run<W, 5>(), last ? "" : ",");
}
int main()
{
    printf("{\"benchmark\": \"tools/ubench/mfma_loop_model.hip\", \"unit\": \"11 ds_read_b128 + 12 v_mfma_f32_32x32x16_f16 + 48 tracking VALU per wave\", \"rows\": [\n");
    row<4>(false); row<8>(false); row<12>(true);
    printf("]}\n");
    return 0;
}
