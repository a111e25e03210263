#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
int main(){
  const float y = 0.04f; long bad = 0; uint32_t hi; float top = 30.0f; memcpy(&hi,&top,4);
  #pragma omp parallel for reduction(+:bad) schedule(static)
  for (uint32_t u = 0; u <= hi; ++u) {
    float x; memcpy(&x,&u,4);
    float q0 = x*y; float r = fmaf(-q0, 25.0f, x); float q1 = fmaf(r, y, q0);
    float ref = (float)((double)x/25.0);
    if (q1 != ref) { bad++; }
    float ref2 = x/25.0f; if (ref2 != ref) bad += 1000000;
  }
  printf("bad=%ld of %u\n", bad, hi+1); return 0; }
