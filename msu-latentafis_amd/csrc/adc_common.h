// adc_common.h — small device helpers shared by the ADC translation units (adc.hip: adc_variant 8; adc_direct.hip: the direct kernels, test library only).
#pragma once
#include "afis_device.h"

namespace afis {

__device__ __forceinline__ void wave_argmax(float& v, int& i)
{
    // max value; on equal values the smaller point index (std::max_element returns the FIRST maximum)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        const bool take = (ov > v) | ((ov == v) & (oi < i));     // bitwise: no short-circuit branches
        v = take ? ov : v;
        i = take ? oi : i;
    }
}

template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) { return __int_as_float(dpp_i<CTRL>(__float_as_int(x))); }
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi)
{
    const bool take = (ov > v) | ((ov == v) & (oi < i));
    v = take ? ov : v;
    i = take ? oi : i;
}

}  // namespace afis
