// Micro-benchmark: issue cost of the vector instructions the graph / candidate kernels are made of, on MI355X (gfx950), at 3 and 4 waves per SIMD.
// Per instruction: shader cycles (s_memtime) and wall nanoseconds per wave64 instruction per SIMD (the chip's clock moves with the instruction mix,
// so the wall figure is the one to compare).  8 independent instructions on 8 registers per loop body.
//   hipcc --offload-arch=gfx950 -O3 valu_cost2.hip -o valu_cost2 && ./valu_cost2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define LIST(X) \
    X(0, "v_add_u32 (VOP2)", "v_add_u32 %0, %0, %8") \
    X(1, "v_sub_u32 (VOP2)", "v_sub_u32 %0, %0, %8") \
    X(2, "v_and_b32 (VOP2)", "v_and_b32 %0, %0, %8") \
    X(3, "v_or_b32 (VOP2)", "v_or_b32 %0, %0, %8") \
    X(4, "v_lshlrev_b32 (VOP2)", "v_lshlrev_b32 %0, 1, %0") \
    X(5, "v_ashrrev_i32 (VOP2)", "v_ashrrev_i32 %0, 1, %0") \
    X(6, "v_min_u32 (VOP2)", "v_min_u32 %0, %0, %8") \
    X(7, "v_max_i32 (VOP2)", "v_max_i32 %0, %0, %8") \
    X(8, "v_mul_f32 (VOP2)", "v_mul_f32 %0, %0, %8") \
    X(9, "v_sub_f32 (VOP2)", "v_sub_f32 %0, %0, %8") \
    X(10, "v_mov_b32 (VOP1)", "v_mov_b32 %0, %8") \
    X(11, "v_cndmask_b32 vcc (VOP2 e32)", "v_cndmask_b32 %0, %0, %8, vcc") \
    X(12, "v_cndmask_b32 sgpr pair (VOP3)", "v_cndmask_b32_e64 %0, %0, %8, s[20:21]") \
    X(13, "v_cmp_lt_u32 -> vcc (VOPC e32)", "v_cmp_lt_u32 vcc, %0, %8") \
    X(14, "v_cmp_lt_u32 -> sgpr pair (VOP3)", "v_cmp_lt_u32_e64 s[20:21], %0, %8") \
    X(15, "v_lshl_add_u32 (VOP3)", "v_lshl_add_u32 %0, %0, 1, %8") \
    X(16, "v_lshl_or_b32 (VOP3)", "v_lshl_or_b32 %0, %0, 1, %8") \
    X(17, "v_and_or_b32 (VOP3)", "v_and_or_b32 %0, %0, %8, %9") \
    X(18, "v_bfe_u32 (VOP3)", "v_bfe_u32 %0, %0, 3, 8") \
    X(19, "v_mad_u32_u24 (VOP3)", "v_mad_u32_u24 %0, %0, %8, %9") \
    X(20, "v_mul_u32_u24 (VOP2)", "v_mul_u32_u24 %0, %0, %8") \
    X(21, "v_mul_lo_u32 (VOP3)", "v_mul_lo_u32 %0, %0, %8") \
    X(22, "v_ffbl_b32 (VOP1)", "v_ffbl_b32 %0, %0") \
    X(23, "v_bcnt_u32_b32 (VOP3)", "v_bcnt_u32_b32 %0, %0, %8") \
    X(24, "v_cvt_f32_i32 (VOP1)", "v_cvt_f32_i32 %0, %0") \
    X(25, "v_rsq_f32 (VOP1, transcendental)", "v_rsq_f32 %0, %0") \
    X(26, "v_rcp_f32 (VOP1, transcendental)", "v_rcp_f32 %0, %0") \
    X(27, "v_dot2_i32_i16 (VOP3P)", "v_dot2_i32_i16 %0, %0, %8, 0") \
    X(28, "v_fmac_f32 (VOP2)", "v_fmac_f32 %0, %8, %9") \
    X(29, "v_max_f32 (VOP2)", "v_max_f32 %0, %0, %8") \
    X(30, "v_max3_f32 (VOP3)", "v_max3_f32 %0, %0, %8, %9") \
    X(31, "v_cmp_gt_u64 -> vcc (VOPC e32)", "v_cmp_gt_u64 vcc, %10, %11") \
    X(32, "v_addc_co_u32 vcc (VOP2)", "v_addc_co_u32 %0, vcc, 0, %0, vcc") \
    X(33, "v_mbcnt_lo_u32_b32 (VOP3)", "v_mbcnt_lo_u32_b32 %0, %8, %0") \
    X(34, "v_add_f32 (VOP2)", "v_add_f32 %0, %0, %8") \
    X(35, "v_pk_sub_i16 (VOP3P)", "v_pk_sub_i16 %0, %0, %8") \
    X(36, "v_xad_u32 (VOP3)", "v_xad_u32 %0, %0, %8, %9") \
    X(37, "v_sub_co_u32 -> vcc (VOP2)", "v_sub_co_u32 %0, vcc, %0, %8") \
    X(38, "v_cndmask_b32_e64 with vcc (VOP3)", "v_cndmask_b32_e64 %0, %0, %8, vcc") \
    X(39, "v_cndmask_b32 vcc, dst not a source (VOP2 e32)", "v_cndmask_b32 %0, %8, %9, vcc") \
    X(40, "v_cndmask_b32 vcc, src0 = 0 (VOP2 e32)", "v_cndmask_b32 %0, 0, %0, vcc") \
    X(41, "v_cmp_ne_u32 vcc + v_cndmask_b32 vcc (pair)", "v_cmp_ne_u32 vcc, %0, %8\n v_cndmask_b32 %0, 0, %0, vcc") \
    X(42, "v_cmp_ne_u32 sgpr + v_cndmask_b32_e64 sgpr (pair)", "v_cmp_ne_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, 0, %0, s[20:21]") \
    X(43, "v_sub_u32 + v_min_u32 (pair: wrap without a select)", "v_sub_u32 %1, %0, %8\n v_min_u32 %0, %0, %1") \
    X(44, "v_dot2c_f32_f16 (VOP2)", "v_dot2c_f32_f16 %0, %8, %9") \
    X(45, "v_dot2_f32_f16 (VOP3P)", "v_dot2_f32_f16 %0, %8, %9, 0") \
    X(46, "v_pk_add_f16 (VOP3P)", "v_pk_add_f16 %0, %0, %8") \
    X(47, "v_fma_f32 (VOP3)", "v_fma_f32 %0, %0, %8, %9") \
    X(48, "v_fmaak_f32 (VOP2 + literal)", "v_fmaak_f32 %0, %0, %8, 0x3f000000") \
    X(49, "v_mul_f32 with abs modifier (VOP3)", "v_mul_f32_e64 %0, %0, |%8|") \
    X(50, "v_xor_b32 (VOP2)", "v_xor_b32 %0, %0, %8") \
    X(51, "v_sqrt_f32 (VOP1, transcendental)", "v_sqrt_f32 %0, %0") \
    X(52, "v_cmp_gt_f32 -> sgpr pair (VOP3)", "v_cmp_gt_f32_e64 s[20:21], %0, %8") \
    X(53, "v_bfi_b32 (VOP3)", "v_bfi_b32 %0, %8, %0, %9") \
    X(54, "v_add3_u32 (VOP3)", "v_add3_u32 %0, %0, %8, %9") \
    X(55, "v_lshrrev_b32 (VOP2)", "v_lshrrev_b32 %0, 1, %0") \
    X(56, "v_add_u32 with literal (VOP2 + literal)", "v_add_u32 %0, 0x12345, %0") \
    X(57, "v_readfirstlane_b32 (VOP1 -> SGPR)", "v_readfirstlane_b32 s22, %0")
constexpr int NK = 58;
#define NAME(i, n, a) n,
static const char* kNames[NK] = {LIST(NAME)};
template <int KIND>
__global__ void k(unsigned* out, unsigned long long* cyc, int iters)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    const unsigned b = 0x00010001u * (threadIdx.x & 7) + 3, c = 0x01020304u;
    unsigned long long w0 = ((unsigned long long)threadIdx.x << 13) | 5, w1 = ((unsigned long long)(threadIdx.x ^ 21) << 13) | 7;
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define ONE(r, tmpl) asm volatile(tmpl : "+v"(a[r]) , "+v"(a[(r + 1) & 7]), "+v"(a[(r + 2) & 7]), "+v"(a[(r + 3) & 7]), "+v"(a[(r + 4) & 7]), "+v"(a[(r + 5) & 7]), "+v"(a[(r + 6) & 7]), "+v"(a[(r + 7) & 7]) : "v"(b), "v"(c), "v"(w0), "v"(w1) : "vcc", "s20", "s21", "s22");
#define BODY(i, n, tmpl) if (KIND == i) { ONE(0, tmpl) ONE(1, tmpl) ONE(2, tmpl) ONE(3, tmpl) ONE(4, tmpl) ONE(5, tmpl) ONE(6, tmpl) ONE(7, tmpl) }
        LIST(BODY)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int KIND> void run(int wps, double& cyc, double& ns)
{
    const int iters = 20000, threads = 256 * wps, blocks = 256;
    unsigned* d; unsigned long long* dc;
    hipMalloc(&d, (size_t)blocks * threads * 4); hipMalloc(&dc, (size_t)blocks * threads / 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, dc, 200);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, dc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * threads / 64);
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    cyc = (double)h[h.size() / 2] / ((double)iters * 8) / wps;
    ns = ms * 1e6 / ((double)iters * 8 * wps);
    hipFree(d); hipFree(dc);
}
template <int KIND> void all()
{
    double c3, n3, c4, n4;
    run<KIND>(3, c3, n3); run<KIND>(4, c4, n4);
    printf("  {\"instruction\": \"%s\", \"cycles_3_waves\": %.2f, \"ns_3_waves\": %.2f, \"cycles_4_waves\": %.2f, \"ns_4_waves\": %.2f}%s\n", kNames[KIND], c3, n3, c4, n4, KIND == NK - 1 ? "" : ",");
    if constexpr (KIND + 1 < NK) all<KIND + 1>();
}
int main()
{
    printf("{\"benchmark\": \"tools/ubench/valu_cost2.hip\", \"device\": \"MI355X gfx950\", \"unit\": \"per wave64 instruction per SIMD: shader cycles (s_memtime) and wall nanoseconds\", \"results\": [\n");
    all<0>();
    printf("]}\n");
    return 0;
}
