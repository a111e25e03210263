// atan2f_libm.h — atan2f as GNU libc 2.35 (the image's libm; sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c, i.e. the fdlibm
// single-precision routines) evaluates it, restated operation by operation in fp32 so that the device produces the bits a CPU
// build of the reference gets at matching/matcher.cpp:1516 and :1524.  That libm routine is NOT correctly rounded (it is off by
// one ulp from the exact value for 16 % of the integer coordinate differences |d| <= 2047), so a correctly rounded device atan2
// would disagree with the CPU there.  Restricted to finite arguments (coordinate differences are small integers).
// Every operation below is a single IEEE fp32 add, multiply or divide; the translation units that include this header are
// compiled with -ffp-contract=off.  tests/test_gpu_parity.py compares it with libm's atan2f exhaustively over [-2047, 2047]^2,
// tools/atan2f_check.c does the same on the host.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define AFIS_HD __host__ __device__ __forceinline__
#else
#define AFIS_HD static inline
#endif

AFIS_HD uint32_t afis_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
AFIS_HD float afis_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// atanf for finite x (fdlibm s_atanf.c): argument reduction to one of four break points, odd/even split polynomial
AFIS_HD float afis_atanf_libm(float x)
{
    const float hi0 = afis_u2f(0x3eed6338u), hi1 = afis_u2f(0x3f490fdau), hi2 = afis_u2f(0x3f7b985eu), hi3 = afis_u2f(0x3fc90fdau);
    const float lo0 = afis_u2f(0x31ac3769u), lo1 = afis_u2f(0x33222168u), lo2 = afis_u2f(0x33140fb4u), lo3 = afis_u2f(0x33a22168u);
    const float a0 = afis_u2f(0x3eaaaaabu), a1 = afis_u2f(0xbe4ccccdu), a2 = afis_u2f(0x3e124925u), a3 = afis_u2f(0xbde38e38u),
                a4 = afis_u2f(0x3dba2e6eu), a5 = afis_u2f(0xbd9d8795u), a6 = afis_u2f(0x3d886b35u), a7 = afis_u2f(0xbd6ef16bu),
                a8 = afis_u2f(0x3d4bda59u), a9 = afis_u2f(0xbd15a221u), a10 = afis_u2f(0x3c8569d7u);
    const uint32_t hx = afis_f2u(x), ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {                               // |x| >= 2^25
        const float r = hi3 + lo3;
        return neg ? -hi3 - lo3 : r;
    }
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix < 0x3ee00000u) {                                // |x| < 0.4375
        if (ix < 0x31000000u) return x;                    // |x| < 2^-29
        id = -1;
    } else {
        x = afis_u2f(ix);                                  // fabsf
        if (ix < 0x3f980000u) {                            // |x| < 1.1875
            if (ix < 0x3f300000u) { id = 0; hi = hi0; lo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; hi = hi1; lo = lo1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { id = 2; hi = hi2; lo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; hi = hi3; lo = lo3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -r : r;
}

// atan2f for finite y, x (fdlibm e_atan2f.c)
AFIS_HD float afis_atan2f_libm(float y, float x)
{
    const float tiny = 1.0e-30f;
    const float pi_o_2 = afis_u2f(0x3fc90fdbu), pi = afis_u2f(0x40490fdbu), pi_lo = afis_u2f(0xb3bbbd2eu);
    const uint32_t hx = afis_f2u(x), hy = afis_f2u(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (hx == 0x3f800000u) return afis_atanf_libm(y);      // x == 1.0
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);      // 2*sign(x) + sign(y)
    if (iy == 0) {                                         // y == 0
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;    // x == 0
    const int k = ((int)iy - (int)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                 // |y/x| > 2^60
    else if ((hx >> 31) && k < -60) z = 0.0f;              // |y|/x < -2^60
    else z = afis_atanf_libm(afis_u2f(afis_f2u(y / x) & 0x7fffffffu));
    switch (m) {
    case 0: return z;
    case 1: return afis_u2f(afis_f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}
