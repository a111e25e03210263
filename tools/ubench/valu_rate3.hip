// Micro-benchmark 3: v_fmac_f32 (VOP2), v_pk_fma_f32, v_pk_add_f32, SDWA byte-extract-and-shift, v_add_u32, v_cmp e32 + cndmask e32 (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {1.0001f, 0.9999f};
    const float b = 1.0001f, c = 0.5f; unsigned w = threadIdx.x * 2654435761u;
    unsigned u0 = 0, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
        else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
        else if (KIND == 3) asm volatile("v_lshlrev_b32_sdwa %0, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
            "v_lshlrev_b32_sdwa %2, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_lshlrev_b32_sdwa %3, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
            "v_lshlrev_b32_sdwa %4, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %5, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
            "v_lshlrev_b32_sdwa %6, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_lshlrev_b32_sdwa %7, 9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
            : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w));
        else if (KIND == 4) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w));
        else asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.y + p3.y + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}
template <int KIND> void run(const char* name, int wps)
{
    float* d; hipMalloc(&d, 256 * 2048 * 4);
    const int iters = 20000, threads = wps <= 4 ? 256 * wps : 1024, blocks = wps <= 4 ? 256 : 256 * (wps / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (2.1 GHz)\n", name, wps, ms * 1e-3 * 2.1e9 / ((double)iters * 8 * wps));
    hipFree(d);
}
int main()
{
    for (int w : {4, 8}) { run<0>("v_fmac_f32 (VOP2)", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_add_f32", w); run<3>("v_lshlrev_b32_sdwa BYTE_n", w); run<4>("v_add_u32", w); run<5>("v_cmp e32 + v_cndmask e32", w); }
    return 0;
}
