/* Host check of msu-latentafis_amd/csrc/atan2f_libm.h against the C library's atan2f, exhaustively over the integer grid
 * [-R, R]^2 (R = 2047 by default) plus a sweep of non-integer arguments:
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/atan2f_check tools/atan2f_check.c -lm && /tmp/atan2f_check [R] */
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include "../msu-latentafis_amd/csrc/atan2f_libm.h"

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 2047;
    long bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n)
    for (int dy = -R; dy <= R; ++dy)
        for (int dx = -R; dx <= R; ++dx) {
            const float a = atan2f((float)dy, (float)dx), b = afis_atan2f_libm((float)dy, (float)dx);
            ++n;
            if (afis_f2u(a) != afis_f2u(b)) { if (bad < 5) printf("mismatch dy=%d dx=%d libm=%a here=%a\n", dy, dx, a, b); ++bad; }
        }
    printf("integer grid [-%d, %d]^2: %ld points, %ld mismatches\n", R, R, n, bad);
    long bad2 = 0, n2 = 0;
#pragma omp parallel for reduction(+ : bad2, n2)
    for (int i = 0; i < 4000; ++i)
        for (int j = 0; j < 4000; ++j) {
            const float y = (float)((i - 2000) * 0.37 + 1e-3 * j), x = (float)((j - 2000) * 0.73 - 1e-3 * i);
            const float a = atan2f(y, x), b = afis_atan2f_libm(y, x);
            ++n2;
            if (afis_f2u(a) != afis_f2u(b)) ++bad2;
        }
    printf("non-integer sweep: %ld points, %ld mismatches\n", n2, bad2);
    return bad || bad2 ? 1 : 0;
}
