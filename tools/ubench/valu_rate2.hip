// Micro-benchmark 2: issue cost of the instruction forms the ADC / graph kernels use, 4 and 8 waves per SIMD (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY8(INS) asm volatile(INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7") \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(mask), "v"(ib));
#define I_ADD(x)    "v_add_f32 " x ", " x ", %8\n"
#define I_FMA(x)    "v_fma_f32 " x ", " x ", %8, %8\n"
#define I_MAX(x)    "v_max_f32 " x ", " x ", %8\n"
#define I_CND64(x)  "v_cndmask_b32_e64 " x ", " x ", %8, %9\n"
#define I_CNDVCC(x) "v_cndmask_b32_e32 " x ", " x ", %8, vcc\n"
#define I_XOR(x)    "v_xor_b32 " x ", " x ", %10\n"
#define I_LSHLADD(x) "v_lshl_add_u32 " x ", " x ", 9, %10\n"
#define I_BFE(x)    "v_bfe_u32 " x ", " x ", 8, 8\n"
#define I_CMP(x)    "v_cmp_gt_f32_e64 s[20:21], " x ", %8\n"
#define I_MOV(x)    "v_mov_b32 " x ", %8\n"
#define I_SUB(x)    "v_sub_f32 " x ", " x ", %8\n"
template <int KIND>
__global__ void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f; const int ib = 32;
    const unsigned long long mask = 0x5555555555555555ull;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { BODY8(I_ADD) } else if (KIND == 1) { BODY8(I_FMA) } else if (KIND == 2) { BODY8(I_MAX) }
        else if (KIND == 3) { BODY8(I_CND64) } else if (KIND == 4) { asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc"); BODY8(I_CNDVCC) }
        else if (KIND == 5) { BODY8(I_XOR) } else if (KIND == 6) { BODY8(I_LSHLADD) } else if (KIND == 7) { BODY8(I_BFE) }
        else if (KIND == 8) { asm volatile(I_CMP("%0") I_CMP("%1") I_CMP("%2") I_CMP("%3") I_CMP("%4") I_CMP("%5") I_CMP("%6") I_CMP("%7")
              :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b) : "s20", "s21"); }
        else if (KIND == 9) { BODY8(I_MOV) } else { BODY8(I_SUB) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char* name, int waves_per_simd)
{
    float* d; hipMalloc(&d, 256 * 2048 * 4);
    const int iters = 20000;
    const int threads = waves_per_simd <= 4 ? 256 * waves_per_simd : 1024, blocks = waves_per_simd <= 4 ? 256 : 256 * (waves_per_simd / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (2.1 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.1e9 / ((double)iters * 8 * waves_per_simd));
    hipFree(d);
}
int main()
{
    for (int w : {4, 8}) {
        run<0>("v_add_f32", w); run<10>("v_sub_f32", w); run<1>("v_fma_f32", w); run<2>("v_max_f32", w); run<3>("v_cndmask_e64 (sgpr)", w); run<4>("v_cndmask_e32 (vcc)", w);
        run<5>("v_xor_b32", w); run<6>("v_lshl_add_u32", w); run<7>("v_bfe_u32", w); run<8>("v_cmp_gt_f32_e64", w); run<9>("v_mov_b32", w);
    }
    return 0;
}
