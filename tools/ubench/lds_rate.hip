// Achievable LDS read bandwidth on gfx950 with the access shape of the ADC kernel: ds_read_b128, every 16-lane group on 16
// distinct 16-byte bank slots (conflict-free), data-dependent addresses, 16 waves per CU, 128 KB of LDS per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip && ./lds_rate
// Prints bytes/s over the whole chip and bytes per CU per clock at the clock rate the runtime reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(float4* out, int iters, int ilp_dummy)
{
    __shared__ float4 lut[8192];                                   // 128 KB
    for (int i = threadIdx.x; i < 8192; i += 1024) lut[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // slot = lane & 15 (distinct within every 16-lane group); the row part of the address changes every step
    uint32_t a = (uint32_t)((threadIdx.x * 97u) & 511u);
    // eight precomputed byte addresses per lane (slot = lane & 15, rows differ); the reads use immediate offsets on top of them
    // (rows +0 .. +15), so the loop is pure LDS issue: no VALU between the reads.  asm volatile keeps every read.
    uint32_t base[4];
    for (int u = 0; u < 4; ++u) base[u] = (((a + 127u * u) & 255u) * 16u + (uint32_t)(lane & 15)) * 16u;
    float4 v0, v1, v2, v3;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("ds_read_b128 %0, %4 offset:%c8\n\tds_read_b128 %1, %5 offset:%c9\n\tds_read_b128 %2, %6 offset:%c10\n\tds_read_b128 %3, %7 offset:%c11"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(base[0]), "v"(base[1]), "v"(base[2]), "v"(base[3]),
                           "n"(u * 256 * 16 & 0xffff), "n"((u * 256 + 256 * 8) * 16 & 0xffff), "n"(u * 256 * 16 + 4096 & 0xffff), "n"(u * 256 * 16 + 8192 & 0xffff));
            if (u & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        keep += v0.x + v1.y + v2.z + v3.w;
    }
    const float4 acc0 = make_float4(keep, 0, 0, 0), acc1 = acc0, acc2 = acc0, acc3 = acc0;
    out[blockIdx.x * 1024 + threadIdx.x] = make_float4(acc0.x, acc1.y, acc2.z, acc3.w);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000, blocks = cus * 4;
    float4* d; hipMalloc(&d, (size_t)blocks * 1024 * sizeof(float4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, 100, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, iters, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 1024 * iters * 32 * 16;
    const double bps = bytes / (ms * 1e-3);
    printf("CUs %d, clock %.0f MHz, %.1f ms: %.1f TB/s LDS reads = %.1f B per CU per clock (at the reported clock); %.3e 4-byte look-ups/s\n",
           cus, p.clockRate / 1e3, ms, bps / 1e12, bps / cus / (p.clockRate * 1e3), bps / 4);
    return 0;
}
