// Is the hardware v_sqrt_f32 (what __fsqrt_rn / __ocml_native_sqrt_f32 compile to) correctly rounded on gfx950?  (No: ~15 % of
// all inputs are off by one ulp.)  And is afis_device.h::sqrt_rn_pos, on its argument range?  (Yes: 0 mismatches.)
// Exhaustive: for every non-negative finite float x, r = v_sqrt_f32(x) is checked against the definition of round-to-nearest:
// ((r_down + r)/2)^2 <= x <= ((r + r_up)/2)^2, evaluated exactly in double (25-bit midpoints, 50-bit squares).
//   hipcc --offload-arch=gfx950 -O3 -o sqrt_exact sqrt_exact.hip && ./sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../msu-latentafis_amd/csrc/afis_device.h"
template <bool FIXED>
__global__ void k(unsigned long long* mism, unsigned* first_bad, unsigned lo, unsigned hi)
{
    unsigned long long local = 0;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const float r = FIXED ? afis::sqrt_rn_pos(x) : __ocml_native_sqrt_f32(x);
        bool ok;
        if (x == 0.0f) ok = (r == 0.0f);
        else {
            const unsigned rb = __float_as_uint(r);
            const double dn = (double)__uint_as_float(rb - 1), up = (double)__uint_as_float(rb + 1), rd = (double)r;
            const double mlo = 0.5 * (dn + rd), mhi = 0.5 * (rd + up);
            ok = (r > 0.0f) && (mlo * mlo <= (double)x) && ((double)x <= mhi * mhi);
        }
        if (!ok) { ++local; atomicMin(first_bad, (unsigned)b); }
    }
    if (local) atomicAdd(mism, local);
}
int main()
{
    unsigned long long* d; unsigned* fb; hipMalloc(&d, 8); hipMalloc(&fb, 4);
    // 1.0f = 0x3f800000, 2^64 = 0x5f800000
    struct { const char* name; bool fixed; unsigned lo, hi; } ranges[] = {
        {"v_sqrt_f32 (= __fsqrt_rn), all normal floats", false, 0x00800000u, 0x7f800000u},
        {"v_sqrt_f32, [1, 2^64)", false, 0x3f800000u, 0x5f800000u},
        {"sqrt_rn_pos, [1, 2^64)", true, 0x3f800000u, 0x5f800000u},
        {"sqrt_rn_pos, zero", true, 0u, 1u}};
    for (auto& r : ranges) {
        unsigned long long z = 0; unsigned big = 0xffffffffu;
        hipMemcpy(d, &z, 8, hipMemcpyHostToDevice); hipMemcpy(fb, &big, 4, hipMemcpyHostToDevice);
        if (r.fixed) hipLaunchKernelGGL(k<true>, dim3(4096), dim3(256), 0, 0, d, fb, r.lo, r.hi);
        else hipLaunchKernelGGL(k<false>, dim3(4096), dim3(256), 0, 0, d, fb, r.lo, r.hi);
        hipDeviceSynchronize();
        hipMemcpy(&z, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&big, fb, 4, hipMemcpyDeviceToHost);
        printf("%s: %llu mismatches of %u inputs (first bad bits 0x%08x)\n", r.name, z, r.hi - r.lo, big);
    }
    return 0;
}
