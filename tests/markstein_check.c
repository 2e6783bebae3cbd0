/* Test infrastructure: proves the division-free mean of the HIP degrid fast path (kbe_device.h,
 * degrid_pixel_fast).  For every float s in [lo, hi] and n = 1..4:
 *     q = s * y;  q' = fma(fma(-2n, q, s), y, q)   with y = RN(1 / (2n))
 * must equal the correctly rounded s / (float) (2n).  Prints the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    const float ys[4] = { 0.5f, 0.25f, 0.16666667f, 0.125f };
    const float lo = argc > 1 ? (float) atof(argv[1]) : 1.0f, hi = argc > 2 ? (float) atof(argv[2]) : 1.7e7f;
    uint32_t a, b;
    long bad = 0, n = 0;
    memcpy(&a, &lo, 4);
    memcpy(&b, &hi, 4);
    for (int k = 1; k <= 4; k++) {
        const float y = ys[k - 1], cf = (float) (2 * k);
        for (uint32_t u = a; u <= b; u++) {
            float s;
            memcpy(&s, &u, 4);
            volatile float ref = s / cf;
            float q = s * y;
            q = fmaf(fmaf(-cf, q, s), y, q);
            bad += q != ref;
            n++;
        }
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
