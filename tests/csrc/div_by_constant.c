/* a / b with y = RN(1/b): q0 = a*y, two FMA-residual corrections; compared with the IEEE quotient.
 * Mirrors div_by() of sydr_amd/csrc/track.hip.  Test infrastructure (tests/test_div_by_constant.py). */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void){ s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv){
  double bs[] = {25e6, 4e6, 50e6, 10e6, 12.5e6, 16.368e6, 3.1415926535898*2.0, 1e-3, 4e-3, 0.25, 0.53, 20e-3, 5.456e6, 38.192e6, 99.375e6};
  for (unsigned k = 0; k < sizeof(bs)/sizeof(bs[0]); ++k) {
    double b = bs[k], y = 1.0 / b;
    long bad3 = 0, bad5 = 0, n = argc > 1 ? atol(argv[1]) : 40000000;
    for (long i = 0; i < n; ++i) {
      uint64_t m = rnd();
      int e = (int)(rnd() % 80) - 40;
      double a = ldexp(1.0 + (double)(m >> 12) / 4503599627370496.0, e);
      if (m & 1) a = -a;
      double q = a / b;
      double q0 = a * y;
      double r0 = fma(-b, q0, a);
      double q1 = fma(r0, y, q0);
      double r1 = fma(-b, q1, a);
      double q2 = fma(r1, y, q1);
      bad3 += q1 != q;
      bad5 += q2 != q;
    }
    printf("b=%.10g  3-op mismatches %ld  5-op mismatches %ld of %ld\n", b, bad3, bad5, n);
  }
  return 0;
}
