/* Test infrastructure: proves the fp32 form of the image-plane position in the HIP projection (kbe_device.h,
 * project_xy).  The reference computes  (float) (((double) ix + 0.5 * W) - 0.5)  (common.py:467-468, double
 * literals); for W >= 2 that equals the single fp32 addition  ix + (float) (0.5 * W - 0.5)  for EVERY fp32 ix:
 * both double operations are exact whenever ix is large enough to matter, and a smaller ix cannot move the sum
 * across an fp32 rounding boundary.  This program sweeps fp32 bit patterns (all of them with stride 1) for the
 * sizes given and also checks the fp32 form of the cull `(double) z >= 0.001` (common.py:453).
 *     centre_offset_check [stride] [W ...]       prints: comparisons mismatches */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t) strtoul(argv[1], 0, 10) : 1u;
    static const int def[] = { 2, 3, 4, 5, 7, 64, 100, 255, 512, 921, 1024, 1919, 2048, 4096, 8191, 16384, 32768 };
    int sizes[64], ns = 0;
    for (int i = 2; i < argc && ns < 64; i++) sizes[ns++] = atoi(argv[i]);
    if (ns == 0) { ns = (int) (sizeof(def) / sizeof(def[0])); memcpy(sizes, def, sizeof(def)); }
    long n = 0, bad = 0;
    for (int k = 0; k < ns; k++) {
        const double half = 0.5 * (double) sizes[k];
        const float c = (float) (half - 0.5);
        if ((double) c != half - 0.5) { printf("offset of %d not representable\n", sizes[k]); return 2; }
        for (uint64_t u = 0; u <= 0xFFFFFFFFull; u += stride) {
            const uint32_t bits = (uint32_t) u;
            float a;
            memcpy(&a, &bits, 4);
            if (!(fabsf(a) <= 3.0e38f)) continue;                   /* finite inputs only (kbe.h) */
            volatile float ref = (float) (((double) a + half) - 0.5);
            volatile float got = a + c;
            float r = ref, g = got;
            n++;
            if (memcmp(&r, &g, 4) != 0) { if (bad < 5) printf("W=%d ix=%a: %a vs %a\n", sizes[k], a, r, g); bad++; }
        }
    }
    for (uint64_t u = 0; u <= 0xFFFFFFFFull; u += stride) {         /* the cull */
        const uint32_t bits = (uint32_t) u;
        float z;
        memcpy(&z, &bits, 4);
        n++;
        if (((double) z >= 0.001) != (z >= 0.001f)) { if (bad < 5) printf("cull z=%a\n", z); bad++; }
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
