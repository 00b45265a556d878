// Exhaustive check: for every finite binary32 x, is  q = x*c; r = fma(-q, y, x); q' = fma(r, c, q)  (c = RN(1/y)) equal to the
// correctly rounded x / y?  y = 9 and y = 3 (csrc/photometric.hip: div9, div3).
//   gcc -O2 -mfma -fopenmp -ffp-contract=off tests/tools/div_const.c -o /tmp/div_const -lm && /tmp/div_const
// All 2^32 bit patterns take ~40 s on 8 cores; -DSTRIDE=61 checks every 61st pattern (tests/test_abi_and_host.py).
// Result: one mismatch per divisor, x = -0 (comes out as +0).
#ifndef STRIDE
#define STRIDE 1
#endif
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static inline float divc(float x, float y, float c) { float q = x * c; float r = fmaf(-q, y, x); return fmaf(r, c, q); }
int main(void) {
  const float ys[2] = {9.f, 3.f};
  for (int k = 0; k < 2; ++k) {
    const float y = ys[k], c = 1.0f / y;
    unsigned long long bad = 0, bad_normal = 0; uint32_t first = 0;
#pragma omp parallel for reduction(+:bad,bad_normal) schedule(static)
    for (long long i = 0; i < (1LL << 32); i += STRIDE) {
      uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
      if (!isfinite(x)) continue;
      float a = divc(x, y, c), b = x / y;
      uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
      if (ua != ub) { ++bad; if (fabsf(x) >= 1e-30f && fabsf(x) <= 1e30f) { ++bad_normal; first = u; } }
    }
    printf("y = %g: %llu mismatches, %llu of them with 1e-30 <= |x| <= 1e30 (one such x: 0x%08x)\n", y, bad, bad_normal, first);
    if (bad_normal != 0 || bad > 1) return 1;
  }
  return 0;
}
