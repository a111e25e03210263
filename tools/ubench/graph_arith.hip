// Exhaustive / adversarial checks of the cheaper arithmetic the graph kernels use on their packed paths (graph.hip):
//   (1) sqrt_rn_int: RN(sqrt(n)) for integer n via v_rsq_f32 + one fma correction, against sqrt_rn_pos — every n in [0, 2*2047^2];
//   (2) texture compatibility "16 |RN sqrt n1 - RN sqrt n2| < 30" decided without square roots, every (n1, n2) in [0, 4802]^2;
//   (3) minutiae compatibility "|RN sqrt n1 - RN sqrt n2| < 30" decided without square roots outside a guard band: n1 over the
//       whole range, n2 within +-12 of (sqrt n1 +- 30)^2 and random; reports mismatches outside the band and the band's hit rate.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../msu-latentafis_amd/csrc -o graph_arith graph_arith.hip && ./graph_arith
#include <hip/hip_runtime.h>
#include <cstdio>
#include "graph_arith.h"
using afis::sqrt_rn_pos; using afis::sqrt_rn_int;
template <int TEX> __device__ __forceinline__ int pair_alg(float f1, float f2) { return afis::pair_compatible_alg<TEX != 0>(f1, f2); }

__global__ void k_sqrt(unsigned n_max, unsigned long long* bad, unsigned* first_bad)
{
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    if (n > n_max) return;
    const float x = (float)n;
    if (__float_as_uint(sqrt_rn_int(x)) != __float_as_uint(sqrt_rn_pos(x))) { atomicAdd(bad, 1ull); atomicMin(first_bad, n); }
}

// (2): thr = 1.875 block units, c = thr^2 = 3.515625; same decision rule as (3)
__global__ void k_tex(unsigned long long* cnt)
{
    const unsigned n1 = blockIdx.x, n2 = blockIdx.y * 256u + threadIdx.x;
    if (n1 > 4802u || n2 > 4802u) return;
    const float f1 = (float)n1, f2 = (float)n2;
    const bool ref = 16.0f * fabsf(sqrt_rn_pos(f1) - sqrt_rn_pos(f2)) < 30.0f;
    const int a = pair_alg<1>(f1, f2);
    atomicAdd(cnt + 0, 1ull);
    if (a == 2) atomicAdd(cnt + 1, 1ull);
    else if ((a == 1) != ref) atomicAdd(cnt + 2, 1ull);
}

// (3): thr = 30 px
__device__ __forceinline__ void minu_check(unsigned n1, long long n2, unsigned long long* cnt)
{
    if (n2 < 0 || n2 > 8380418ll) return;
    const float f1 = (float)n1, f2 = (float)n2;
    const bool ref = fabsf(sqrt_rn_pos(f1) - sqrt_rn_pos(f2)) < 30.0f;
    const int a = pair_alg<0>(f1, f2);
    atomicAdd(cnt + 0, 1ull);
    if (a == 2) atomicAdd(cnt + 1, 1ull);
    else if ((a == 1) != ref) atomicAdd(cnt + 2, 1ull);
}
__global__ void k_minu_adv(unsigned long long* cnt)
{
    const unsigned n1 = blockIdx.x * 256u + threadIdx.x;
    if (n1 > 8380418u) return;
    const double a = sqrt((double)n1);
    for (int sg = -1; sg <= 1; sg += 2) {
        const double b = a + 30.0 * sg;
        if (b < 0) continue;
        const long long c = (long long)(b * b + 0.5);
        for (int j = -12; j <= 12; ++j) minu_check(n1, c + j, cnt);
    }
}
__global__ void k_minu_rand(unsigned long long* cnt, int max_coord)
{
    unsigned long long st = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    for (int i = 0; i < 4096; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const int dx1 = (int)((st >> 11) % (unsigned)(2 * max_coord + 1)) - max_coord, dy1 = (int)((st >> 33) % (unsigned)(2 * max_coord + 1)) - max_coord;
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const int ex = (int)((st >> 11) % 81u) - 40, ey = (int)((st >> 33) % 81u) - 40;      // the rolled pair differs by a small vector: near-threshold cases
        const int dx2 = dx1 + ex, dy2 = dy1 + ey;
        if (abs(dx2) > 2047 || abs(dy2) > 2047) continue;
        minu_check((unsigned)(dx1 * dx1 + dy1 * dy1), (long long)dx2 * dx2 + (long long)dy2 * dy2, cnt);
    }
}

int main()
{
    unsigned long long *d, h[8] = {}; unsigned *df, hf = 0xffffffffu;
    hipMalloc(&d, 64); hipMalloc(&df, 8);
    const unsigned N = 8380418u;
    hipMemset(d, 0, 64); hipMemcpy(df, &hf, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sqrt, dim3((N + 256) / 256), dim3(256), 0, 0, N, d, df);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, df, 4, hipMemcpyDeviceToHost);
    printf("(1) rsq + fma square root != correctly rounded: %llu of the integers 0..%u (first: %u)\n", h[0], N, hf);

    hipMemset(d, 0, 64);
    hipLaunchKernelGGL(k_tex, dim3(4803, 19), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("(2) texture predicate, every pair of [0, 4802]^2: %llu pairs, %llu in the guard band (%.2e), %llu mismatches outside it\n", h[0], h[1], (double)h[1] / h[0], h[2]);

    hipMemset(d, 0, 64);
    hipLaunchKernelGGL(k_minu_adv, dim3((N + 256) / 256), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("(3) minutiae predicate, near-threshold sweep: %llu pairs, %llu in the guard band, %llu mismatches outside it\n", h[0], h[1], h[2]);
    for (int mc : {60, 300, 2047}) {
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(k_minu_rand, dim3(4096), dim3(256), 0, 0, d, mc);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("(3) minutiae predicate, random pairs (|d| <= %d): %llu pairs, %llu in the guard band (%.2e), %llu mismatches outside it\n", mc, h[0], h[1], (double)h[1] / h[0], h[2]);
    }
    return 0;
}
