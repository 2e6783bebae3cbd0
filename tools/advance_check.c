// CPU check of the integer-mantissa walk of the hole fill (struct Axis, axis_jump, axis_catch_up and advance_exact in
// ken-burns-effect_amd/csrc/kbe_holes.hip, restated here in C): sequences of m fp32 additions taken as the kernel takes
// them against the additions one at a time -- the 16 fill directions, both axes, both senses, starts 0..9000,
// m up to 250 (dev aid).
//   gcc -O2 -ffp-contract=off -o /tmp/advance_check tools/advance_check.c -lm && /tmp/advance_check
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static float asf(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t asu(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
typedef struct { int A, step, e; } Axis;
static int interior(int A, int mag) { return (unsigned) (A - (1 << 23) - mag) < (unsigned) ((1 << 23) - 2 * mag); }
static Axis axis_enter(float f, float u, int subtract)
{
    const uint32_t bits = asu(f);
    const int e = (int) (bits >> 23) - 127;
    if (f >= 32.0f && e <= 22) {
        const float sc = ldexpf(u, 23 - e), r = rintf(sc);
        const int step = subtract ? -(int) r : (int) r;
        const int A = (int) ((bits & 0x7FFFFFu) | 0x800000u);
        if (fabsf(sc - r) != 0.5f && interior(A, abs(step))) return (Axis){ A, step, e };
    }
    return (Axis){ (int) bits, 0, -1 };
}
static float axis_value(Axis ax) { return ax.e >= 0 ? asf(((uint32_t) (ax.e + 127) << 23) | ((uint32_t) ax.A & 0x7FFFFFu)) : asf((uint32_t) ax.A); }
static int axis_pixel(Axis ax) { if (ax.e >= 0) { int sh = 23 - ax.e; return (ax.A + (1 << (sh - 1))) >> sh; } return (int) roundf(asf((uint32_t) ax.A)); }
static void axis_jump(Axis* ax, int* r) { if (ax->e >= 0) { int end = ax->A + *r * ax->step; if (interior(end, abs(ax->step))) { ax->A = end; *r = 0; } } }
static long n_catch = 0;
static void axis_catch_up(Axis* ax, int* r, float u, int subtract, float limit)
{
    n_catch++;
    if (u == 0.0f) { *r = 0; return; }
    if (ax->e >= 0) {
        const int mag = abs(ax->step) > 1 ? abs(ax->step) : 1;
        const int room = ax->step < 0 ? ax->A - (1 << 23) - mag : (1 << 24) - 1 - mag - ax->A;
        int j = (int) ((float) room * (1.0f / (float) mag) * 1.000001f) - 1;      // a sloppier reciprocal than v_rcp_f32
        if (j > *r) j = *r;
        if (j >= 1 && interior(ax->A + j * ax->step, mag)) { ax->A += j * ax->step; *r -= j; }
    }
    volatile float f = axis_value(*ax);
    for (int i = 0; i < 4; i++) if (*r > 0) { f = subtract ? f - u : f + u; (*r)--; }
    if (f < -1.0f || f > limit) *r = 0;
    *ax = axis_enter(f, u, subtract);
    if (*r > 0) axis_jump(ax, r);
}
static float advance_exact(float a0, float u, int m, int subtract)
{
    volatile float a = a0;
    if (u == 0.0f) return a0;
    while (m > 0) {
        const uint32_t bits = asu(a); const int e = (int) (bits >> 23) - 127;
        if (m >= 3 && a >= 1.0f && e <= 23) {
            const float s = ldexpf(u, 23 - e), r = rintf(s);
            if (fabsf(s - r) != 0.5f) {
                const int step = (int) r, mag = abs(step), A = (int) ((bits & 0x7FFFFFu) | 0x800000u);
                const int down = subtract ? step > 0 : step < 0;
                const int room = down ? A - (1 << 23) - mag : (1 << 24) - 1 - mag - A, other = down ? (1 << 24) - 1 - mag - A : A - (1 << 23) - mag;
                int j = (room > 0 && other >= 0 && mag > 0) ? (int) ((float) room / (float) mag) - 1 : 0; if (j > m) j = m;
                if (j >= 1) { const int end = A + j * (down ? -mag : mag); a = asf((bits & 0xFF800000u) | ((uint32_t) end & 0x7FFFFFu)); m -= j; continue; }
            }
        }
        a = subtract ? a - u : a + u; m--;
    }
    return a;
}
int main(void)
{
    const float dx[16] = { -1, 0, 1, 1, -1, 1, 2, 2, -2, -1, 1, 2, 3, 3, 3, 3 }, dy[16] = { 1, 1, 1, 0, 2, 2, 1, -1, 3, 3, 3, 3, 2, 1, -1, -2 };
    long walks = 0, advances = 0, bad = 0; uint32_t seed = 12345;
    for (int d = 0; d < 16; d++) {
        volatile float n = sqrtf(dx[d] * dx[d] + dy[d] * dy[d]); const float us[2] = { dx[d] / n, dy[d] / n };
        for (int ax = 0; ax < 2; ax++) for (int sub = 0; sub < 2; sub++) for (int start = 0; start < 9000; start += (start < 1100 ? 1 : 7)) {
            const float u = us[ax];
            Axis A = axis_enter((float) start, u, sub);
            volatile float seq = (float) start; int total = 0;
            walks++;
            for (int leg = 0; leg < 12; leg++) {
                const int m = (int) ((seed = seed * 1664525u + 1013904223u) >> 8) % ((leg & 1) ? 250 : 9) + 1;
                for (int k = 0; k < m; k++) seq = sub ? seq - u : seq + u;
                total += m;
                int r = m; axis_jump(&A, &r);
                for (int guard = 0; r > 0; guard++) { axis_catch_up(&A, &r, u, sub, 20000.0f); if (guard > 1000) { printf("stuck\n"); return 1; } }
                advances++;
                if (seq < -1.0f || seq > 20000.0f) break;           // the kernel stops caring here
                const float got = axis_value(A), ex = advance_exact((float) start, u, total, sub);
                if (asu(got) != asu((float) seq) || asu(ex) != asu((float) seq) || axis_pixel(A) != (int) roundf(seq)) {
                    if (bad++ < 5) printf("MISMATCH d=%d axis=%d sub=%d start=%d after %d steps: axis %.9g, advance_exact %.9g, one at a time %.9g\n", d, ax, sub, start, total, (double) got, (double) ex, (double) seq);
                }
            }
        }
    }
    printf("walks %ld, advances %ld (catch-ups %ld), mismatches %ld\n", walks, advances, n_catch, bad);
    return bad != 0;
}
