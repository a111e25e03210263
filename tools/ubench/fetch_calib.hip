// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ counters on gfx950 for the access patterns of this repository, on KNOWN byte counts.
//   mode 0  stream : every lane reads 16 B, fully coalesced, over a buffer of `mb` MiB, once            -> known bytes = buffer size
//   mode 1  gather4: every lane reads 4 B at a hashed (pseudo-random) dword of a table of `mb` MiB, `n` gathers in total, 16 independent
//                    gathers per lane in flight (the refine of k_adc_rowmin_q: 16 dependent-free look-ups per candidate)
//                    table >> L2 + Infinity Cache (e.g. 16384 MiB): nearly every gather misses everything -> one fabric request each
//                    table = 86 MiB (one query group's fp32 table): the Infinity Cache absorbs most of them
//   mode 2  stream8: every lane reads 8 B, fully coalesced (the recomputation kernel's record stream: 512 contiguous bytes per wave load), once       -> known bytes = buffer size
// Run each mode under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  and under  --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ; this program prints the
// known counts as JSON, tools/profile_round3.sh joins them with the counters.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib && ./fetch_calib <mode> <mb> [n_gathers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_stream(const uint4* __restrict__ p, size_t n16, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream8(const uint2* __restrict__ p, size_t n8, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) { const uint2 v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_gather4(const unsigned* __restrict__ t, unsigned long long n_dwords, unsigned long long n_gathers, unsigned* out)
{
    unsigned acc = 0;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = tid * 16; g < n_gathers; g += nthr * 16) {
        unsigned v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned long long h = ((unsigned long long)hash((unsigned)(g + j)) << 32 | hash((unsigned)((g + j) >> 32) ^ 0x9e3779b9u ^ hash((unsigned)(g + j) + 77u)));
            v[j] = t[h % n_dwords];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t mb = argc > 2 ? (size_t)atoll(argv[2]) : 4096;
    const unsigned long long n = argc > 3 ? strtoull(argv[3], nullptr, 10) : (1ull << 28);
    const size_t bytes = mb << 20;
    void* d = nullptr; unsigned* out = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(d, 1, bytes); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out);
    else if (mode == 2) hipLaunchKernelGGL(k_stream8, dim3(256 * 16), dim3(256), 0, 0, (const uint2*)d, bytes / 8, out);
    else hipLaunchKernelGGL(k_gather4, dim3(256 * 16), dim3(256), 0, 0, (const unsigned*)d, (unsigned long long)(bytes / 4), n, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (mode == 2) printf("{\"mode\": \"stream8\", \"kernel\": \"k_stream8\", \"buffer_mib\": %zu, \"known_bytes\": %zu, \"ms\": %.3f, \"GBps\": %.1f}\n", mb, bytes, ms, bytes / (ms * 1e-3) / 1e9);
    else if (mode == 0) printf("{\"mode\": \"stream16\", \"kernel\": \"k_stream\", \"buffer_mib\": %zu, \"known_bytes\": %zu, \"ms\": %.3f, \"GBps\": %.1f}\n", mb, bytes, ms, bytes / (ms * 1e-3) / 1e9);
    else printf("{\"mode\": \"gather4\", \"kernel\": \"k_gather4\", \"table_mib\": %zu, \"gathers\": %llu, \"useful_bytes\": %llu, \"bytes_if_64B_per_gather\": %llu, \"ms\": %.3f, \"gathers_per_s\": %.3e}\n",
                mb, n, n * 4ull, n * 64ull, ms, n / (ms * 1e-3));
    return 0;
}
