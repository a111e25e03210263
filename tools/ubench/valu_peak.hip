// Micro-benchmark: VALU issue cost per wave64 instruction on MI355X (gfx950), by encoding / opcode class, at 1, 2, 3, 4, 6 and 8 waves per SIMD.
// Cycles are SHADER cycles read in the kernel (s_memtime), not wall time divided by an assumed clock; the effective clock (cycles / wall) is
// printed beside them.  One workgroup per CU (256 blocks x 256*w threads: w waves on each of the 4 SIMDs; w = 8 uses two 1024-thread blocks per CU).
// Every instruction group is 8 independent instructions on 8 different registers, so dependent-issue latency does not enter.
//   hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak && ./valu_peak > profiles/r03_valu_peak.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
enum { K_ADD_F32, K_ADD_U32, K_ADD3_U32, K_PK_ADD_U16, K_PK_MIN_U16, K_PK_SUB_I16, K_PK_ASHR_I16, K_BFI, K_PERM, K_MAX3_F32, K_MED3_F32, K_MAX_F32, K_MOV_DPP,
       K_CNDMASK_VCC, K_FMA_F32, K_PK_ADD_F32, K_XOR, K_MIN3_U32, K_NKINDS };
static const char* kNames[K_NKINDS] = {"v_add_f32 (VOP2 e32)", "v_add_u32 (VOP2 e32)", "v_add3_u32 (VOP3)", "v_pk_add_u16 (VOP3P)", "v_pk_min_u16 (VOP3P)", "v_pk_sub_i16 (VOP3P)",
    "v_pk_ashrrev_i16 (VOP3P)", "v_bfi_b32 (VOP3)", "v_perm_b32 (VOP3)", "v_max3_f32 (VOP3)", "v_med3_f32 (VOP3)", "v_max_f32 (VOP2 e32)", "v_mov_b32 dpp quad_perm (VOP1 DPP)",
    "v_cndmask_b32 vcc (VOP2 e32)", "v_fma_f32 (VOP3)", "v_pk_add_f32 (VOP3P)", "v_xor_b32 (VOP2 e32)", "v_min3_u32 (VOP3)"};

template <int KIND>
__global__ void k(unsigned* out, unsigned long long* cyc, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned b = 0x00010001u * (threadIdx.x & 7), c = 0x01020304u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define OPS(fmt) asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc")
#define F_ADD_F32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define F_ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define F_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define F_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define F_PKMIN(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n"
#define F_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define F_PKASHR(n) "v_pk_ashrrev_i16 %" #n ", 15, %" #n " op_sel_hi:[0,1]\n"
#define F_BFI(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n"
#define F_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define F_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define F_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define F_MAX(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define F_DPP(n) "v_mov_b32_dpp %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define F_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define F_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define F_MIN3U(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
        if (KIND == K_ADD_F32) OPS(F_ADD_F32);
        else if (KIND == K_ADD_U32) OPS(F_ADD_U32);
        else if (KIND == K_ADD3_U32) OPS(F_ADD3);
        else if (KIND == K_PK_ADD_U16) OPS(F_PKADD);
        else if (KIND == K_PK_MIN_U16) OPS(F_PKMIN);
        else if (KIND == K_PK_SUB_I16) OPS(F_PKSUB);
        else if (KIND == K_PK_ASHR_I16) OPS(F_PKASHR);
        else if (KIND == K_BFI) OPS(F_BFI);
        else if (KIND == K_PERM) OPS(F_PERM);
        else if (KIND == K_MAX3_F32) OPS(F_MAX3);
        else if (KIND == K_MED3_F32) OPS(F_MED3);
        else if (KIND == K_MAX_F32) OPS(F_MAX);
        else if (KIND == K_MOV_DPP) OPS(F_DPP);
        else if (KIND == K_CNDMASK_VCC) OPS(F_CND);
        else if (KIND == K_FMA_F32) OPS(F_FMA);
        else if (KIND == K_XOR) OPS(F_XOR);
        else if (KIND == K_MIN3_U32) OPS(F_MIN3U);
        else {   // K_PK_ADD_F32: register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {__uint_as_float(a0), __uint_as_float(a1)}, p1 = {__uint_as_float(a2), __uint_as_float(a3)}, p2 = {__uint_as_float(a4), __uint_as_float(a5)}, p3 = {__uint_as_float(a6), __uint_as_float(a7)}, bb = {1.0f, 2.0f};
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
            a0 = __float_as_uint(p0.x); a1 = __float_as_uint(p0.y); a2 = __float_as_uint(p1.x); a3 = __float_as_uint(p1.y); a4 = __float_as_uint(p2.x); a5 = __float_as_uint(p2.y); a6 = __float_as_uint(p3.x); a7 = __float_as_uint(p3.y);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

struct Res { double cyc_per_instr_simd, wall_cyc_24, clock_ghz; };
template <int KIND> Res run(int wps)
{
    const int iters = 20000, threads = wps <= 4 ? 256 * wps : (wps == 6 ? 768 : 1024), blocks = wps <= 4 ? 256 : (wps == 6 ? 512 : 256 * (wps / 4));
    unsigned* d; unsigned long long* dc;
    hipMalloc(&d, (size_t)blocks * threads * 4); hipMalloc(&dc, (size_t)blocks * threads / 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, dc, 200);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, dc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * threads / 64);
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];                            // shader cycles one wave spent on iters x 8 instructions
    Res r;
    r.cyc_per_instr_simd = med / ((double)iters * 8) / wps;                // wps waves share the SIMD: issue interval per SIMD
    r.wall_cyc_24 = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps);
    r.clock_ghz = med / (ms * 1e-3) / 1e9;                                 // every wave runs the whole kernel: cycles / wall = clock
    hipFree(d); hipFree(dc);
    return r;
}
template <int KIND> void all(bool last)
{
    printf("  {\"instruction\": \"%s\", \"cycles_per_wave64_instruction_per_simd\": {", kNames[KIND]);
    const int ws[6] = {1, 2, 3, 4, 6, 8};
    double clk = 0;
    for (int i = 0; i < 6; ++i) { Res r = run<KIND>(ws[i]); clk = r.clock_ghz; printf("\"%d_waves\": %.3f%s", ws[i], r.cyc_per_instr_simd, i < 5 ? ", " : ""); }
    printf("}, \"effective_clock_ghz_at_8_waves\": %.3f}%s\n", clk, last ? "" : ",");
}
int main()
{
    printf("{\"benchmark\": \"tools/ubench/valu_peak.hip\", \"device\": \"MI355X gfx950\", \"method\": \"8 independent instructions per loop body, 20000 iterations, one workgroup of w waves per SIMD on each of the 256 CUs; cycles = median over waves of s_memtime deltas / (instructions x w)\", \"results\": [\n");
    all<K_ADD_F32>(false); all<K_ADD_U32>(false); all<K_XOR>(false); all<K_MAX_F32>(false); all<K_CNDMASK_VCC>(false); all<K_MOV_DPP>(false);
    all<K_ADD3_U32>(false); all<K_BFI>(false); all<K_PERM>(false); all<K_FMA_F32>(false); all<K_MAX3_F32>(false); all<K_MED3_F32>(false); all<K_MIN3_U32>(false);
    all<K_PK_ADD_U16>(false); all<K_PK_MIN_U16>(false); all<K_PK_SUB_I16>(false); all<K_PK_ASHR_I16>(false); all<K_PK_ADD_F32>(true);
    printf("]}\n");
    return 0;
}
