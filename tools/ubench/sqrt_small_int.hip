// Is the hardware v_sqrt_f32 already correctly rounded on the integer arguments the graph kernels feed it?
//   texture lists: n = dx^2 + dy^2 in block units, n <= 4802;  minutiae lists (packed path): n <= 2 * 2047^2 = 8 380 418
// Prints the number of integers in [0, N] whose hardware square root differs from the correctly rounded one (sqrt_rn_pos).
//   hipcc --offload-arch=gfx950 -O3 -I../../msu-latentafis_amd/csrc -o sqrt_small_int sqrt_small_int.hip && ./sqrt_small_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include "afis_device.h"
__global__ void k(unsigned n_max, unsigned long long* bad_small, unsigned long long* bad_all, unsigned* first_bad)
{
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    if (n > n_max) return;
    const float x = (float)n;
    const float hw = __builtin_amdgcn_sqrtf(x), rn = afis::sqrt_rn_pos(x);
    if (n == 0 ? hw != 0.0f : hw != rn) { atomicAdd(bad_all, 1ull); if (n <= 4802u) atomicAdd(bad_small, 1ull); atomicMin(first_bad, n); }
}
int main()
{
    unsigned long long *d, h[2] = {0, 0}; unsigned *df, hf = 0xffffffffu;
    hipMalloc(&d, 16); hipMalloc(&df, 4); hipMemcpy(d, h, 16, hipMemcpyHostToDevice); hipMemcpy(df, &hf, 4, hipMemcpyHostToDevice);
    const unsigned N = 8380418u;
    hipLaunchKernelGGL(k, dim3((N + 256) / 256), dim3(256), 0, 0, N, d, d + 1, df);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, df, 4, hipMemcpyDeviceToHost);
    printf("hardware sqrt != correctly rounded: %llu of the integers 0..4802, %llu of 0..%u (first: %u)\n", h[0], h[1], N, hf);
    return 0;
}
