// Checks the operand / result lane maps of v_mfma_f32_32x32x16_f16 on the device against the maps adc_mfma.hip assumes:
//   A (M x K = 32 x 16): lane l holds A[i = l & 31][k = 8 * (l >> 5) + e], e = 0..7
//   B (K x N = 16 x 32): lane l holds B[k = 8 * (l >> 5) + e][j = l & 31]
//   C/D (32 x 32): lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
// with asymmetric integer-valued A and B (exact in f16 / f32), and measures the issue rate with 4 independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k_check(float* out)
{
    const int l = threadIdx.x;
    half8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int i = l & 31, k = 8 * (l >> 5) + e, j = l & 31;
        a[e] = (_Float16)(float)((i * 3 + k * 5) % 11 - 5);          // A[i][k]
        b[e] = (_Float16)(float)((k * 7 + j * 2) % 13 - 6);          // B[k][j]
    }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = (float)(r + 100 * (l >> 5));   // C[row][col] = r + 100 h: only to see it is added
    const floatx16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = d[r];
}
__global__ void k_rate(float* out, int iters)
{
    half8 a, b; floatx16 c0, c1, c2, c3;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 1; c2[r] = 2; c3[r] = 3; }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; (void)hipMalloc(&d, 64 * 16 * 4 + 256 * 1024 * 4);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 16]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            float want = (float)(r + 100 * (l >> 5));
            for (int k = 0; k < 16; ++k) want += (float)((row * 3 + k * 5) % 11 - 5) * (float)((k * 7 + col * 2) % 13 - 6);
            if (h[l * 16 + r] != want) { if (bad < 5) printf("mismatch lane %d reg %d: got %g want %g\n", l, r, h[l * 16 + r], want); ++bad; }
        }
    printf("{\"check\": \"v_mfma_f32_32x32x16_f16 lane maps\", \"mismatches\": %d", bad);
    for (int w : {1, 2}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 20000;
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(256 * w), 0, 0, d + 1024, 100);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_rate, dim3(256), dim3(256 * w), 0, 0, d + 1024, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 4 * w * iters * 4 * 2.0 * 32 * 32 * 16;
        printf(", \"tflops_%d_wave_per_simd\": %.1f", w, flop / (ms * 1e-3) / 1e12);
    }
    printf("}\n");
    return bad != 0;
}
