/* Test infrastructure: the normalisation of k_tiles divides four accumulators by the same weight sum.  With the
 * correctly rounded reciprocal y = RN(1 / b) (one IEEE division) each quotient is
 *     q = a * y;  q' = fma(fma(-b, q, a), y, q)
 * which is the correctly rounded a / b (Markstein) as long as nothing underflows.  This program compares it with
 * the division for random (a, b) with b in [2^-24, 2^14] (the weight sum + 1e-7) and |a| in [2^-60, 2^40], zeros,
 * both signs, and quotients near simple fractions.  Prints "<pairs> <mismatches>". */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline float frand(float lo_exp, float hi_exp) {   /* log-uniform magnitude, random mantissa */
    uint32_t e = (uint32_t) (lo_exp + 127) + (uint32_t) (rnd() % (uint32_t) (hi_exp - lo_exp + 1));
    uint32_t u = (e << 23) | (uint32_t) (rnd() & 0x7FFFFF);
    float f; memcpy(&f, &u, 4); return f;
}
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 200000000L, bad = 0;
    for (long i = 0; i < n; i++) {
        const float b = frand(-24, 14);                 /* den = w + 1e-7: [2^-24, 2^14] */
        float a = frand(-60, 40);
        if ((i & 15) == 0) a = b * (float) (rnd() % 4096) / 7.0f;      /* quotients near simple fractions */
        if ((i & 1023) == 0) a = 0.0f;
        if (i & 1) a = -a;
        volatile float y = 1.0f / b;
        float q = a * y;
        const float r = fmaf(-b, q, a);
        q = fmaf(r, y, q);
        volatile float ref = a / b;
        if (q != ref) { if (bad < 5) printf("a=%a b=%a ref=%a got=%a\n", a, b, ref, q); bad++; }
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
