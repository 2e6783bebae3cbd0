/* kbe_jpeg.c -- libkbe_jpeg.so: the frame writers' JPEG encoder (include/kbe_jpeg.h).  HOST code, plain C, no GPU.
 *
 * Where it sits: the reference hands its finished frames to moviepy -> ffmpeg (`mpeg4`, /root/reference/utils/pipeline.py:130-134).
 * Without an ffmpeg binary this package writes the video itself as Motion-JPEG (pipeline.write_mjpeg_mp4 / _avi), and until
 * round 6 Pillow encoded the frames -- one at a time whatever the thread count (its encoder holds the interpreter lock):
 * 137 ms for the 127 frames of a 512 x 512 video whose three networks and 64 rendered frames take 19 ms.  The frames of a
 * Motion-JPEG stream are independent: this encoder takes a batch of them and spreads it over host threads.
 *
 * What it writes: baseline sequential DCT JPEG (ISO/IEC 10918-1), 8 bits, YCbCr (JFIF 1.01 conversion) with 4:2:0 chroma, the
 * example quantisation tables of Annex K.1 scaled by the IJG quality rule, the typical Huffman tables of Annex K.3 -- the
 * same choices, table for table, as Pillow's default `save(format='JPEG', quality=q)` (tests/test_jpeg_writer.py reads both
 * files' DQT / DHT segments and compares; it decodes this encoder's output with Pillow and holds it against the source).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kbe_jpeg.h"

/* the hot functions are compiled twice, for AVX2 + FMA and for the baseline ISA, and picked at load time (the machine that builds the
 * library is not the machine that runs it: no -march=native) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define KBE_HOT __attribute__((target_clones("avx2,fma", "default")))
#else
#define KBE_HOT
#endif

static const uint8_t ZIGZAG[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
/* ... of the TRANSPOSED block (what fdct leaves): entry i of the scan sits at [u][v] instead of [v][u] */
static uint8_t ZIGZAG_T[64];
/* Annex K.1, natural (row-major) order */
static const uint8_t Q_LUMA[64] = { 16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
static const uint8_t Q_CHROMA[64] = { 17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };
/* Annex K.3: code counts per length 1..16, then the symbols in code order */
static const uint8_t DC_LUMA_BITS[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t DC_CHROMA_BITS[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t DC_VALS[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t AC_LUMA_BITS[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t AC_LUMA_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
    0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
    0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
    0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
static const uint8_t AC_CHROMA_BITS[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t AC_CHROMA_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
    0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38,
    0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };

typedef struct { uint16_t code[256]; uint8_t len[256]; } Huff;
typedef struct {
    uint8_t q[2][64];           /* quantisation tables, natural order */
    float rq[2][64];            /* 1 / (q * the AAN scale factors * 8): what a DCT output is multiplied by; TRANSPOSED, as fdct leaves the block */
    Huff dc[2], ac[2];
} Tables;

static void huff_build(const uint8_t* bits, const uint8_t* vals, Huff* h)
{
    memset(h, 0, sizeof(*h));
    unsigned code = 0;
    int k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len - 1]; i++, k++) { h->code[vals[k]] = (uint16_t) code++; h->len[vals[k]] = (uint8_t) len; }
        code <<= 1;
    }
}

static void tables_build(int quality, Tables* t)
{
    static const double aan[8] = { 1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379 };
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;                /* the IJG rule (jpeg_quality_scaling) */
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 64; i++) {
            long v = ((long) (c ? Q_CHROMA[i] : Q_LUMA[i]) * scale + 50) / 100;
            if (v < 1) v = 1;
            if (v > 255) v = 255;                                                       /* baseline: 8-bit entries */
            t->q[c][i] = (uint8_t) v;
            t->rq[c][(i & 7) * 8 + (i >> 3)] = (float) (1.0 / ((double) v * aan[i >> 3] * aan[i & 7] * 8.0));        /* (transposed: fdct) */
        }
    for (int i = 0; i < 64; i++) ZIGZAG_T[i] = (uint8_t) ((ZIGZAG[i] & 7) * 8 + (ZIGZAG[i] >> 3));        /* (idempotent: every call writes the same values) */
    huff_build(DC_LUMA_BITS, DC_VALS, &t->dc[0]);
    huff_build(DC_CHROMA_BITS, DC_VALS, &t->dc[1]);
    huff_build(AC_LUMA_BITS, AC_LUMA_VALS, &t->ac[0]);
    huff_build(AC_CHROMA_BITS, AC_CHROMA_VALS, &t->ac[1]);
}

/* One pass of the Arai-Agui-Nakajima forward DCT down the COLUMNS of an 8 x 8 block, all eight columns side by side (the loops over x
 * are what the compiler turns into 8-wide vector code); outputs scaled by the factors folded into Tables::rq. */
static inline __attribute__((always_inline)) void fdct_columns(float (*d)[8])
{
    float t0[8], t1[8], t2[8], t3[8], t4[8], t5[8], t6[8], t7[8];
    for (int x = 0; x < 8; x++) {
        t0[x] = d[0][x] + d[7][x]; t7[x] = d[0][x] - d[7][x]; t1[x] = d[1][x] + d[6][x]; t6[x] = d[1][x] - d[6][x];
        t2[x] = d[2][x] + d[5][x]; t5[x] = d[2][x] - d[5][x]; t3[x] = d[3][x] + d[4][x]; t4[x] = d[3][x] - d[4][x];
    }
    for (int x = 0; x < 8; x++) {
        const float t10 = t0[x] + t3[x], t13 = t0[x] - t3[x], t11 = t1[x] + t2[x], t12 = t1[x] - t2[x];
        d[0][x] = t10 + t11; d[4][x] = t10 - t11;
        const float z1 = (t12 + t13) * 0.707106781f;
        d[2][x] = t13 + z1; d[6][x] = t13 - z1;
        const float u10 = t4[x] + t5[x], u11 = t5[x] + t6[x], u12 = t6[x] + t7[x];
        const float z5 = (u10 - u12) * 0.382683433f, z2 = 0.541196100f * u10 + z5, z4 = 1.306562965f * u12 + z5, z3 = u11 * 0.707106781f;
        const float z11 = t7[x] + z3, z13 = t7[x] - z3;
        d[5][x] = z13 + z2; d[3][x] = z13 - z2; d[1][x] = z11 + z4; d[7][x] = z11 - z4;
    }
}
/* the 2-D transform: columns, transpose, columns -- the result is the TRANSPOSED coefficient block (out[v][u]); the caller reads it through
 * a transposed zigzag (ZIGZAG_T) and the transposed reciprocal table */
static inline __attribute__((always_inline)) void fdct(float* blk)
{
    float (*d)[8] = (float (*)[8]) blk;
    float t[8][8];
    fdct_columns(d);
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) t[x][y] = d[y][x];
    fdct_columns(t);
    memcpy(blk, t, sizeof(t));
}

typedef struct { uint8_t* p; uint8_t* end; uint64_t acc; int n; int overflow; } Bits;

static inline void put_byte(Bits* b, unsigned v)
{
    if (b->p < b->end) *b->p++ = (uint8_t) v; else b->overflow = 1;
}
/* the entropy-coded segment: bits gather in a 64-bit word and leave four bytes at a time -- in one store when none of the four is 0xFF
 * (which must be followed by a stuffed zero byte, B.1.1.5), byte by byte otherwise or near the end of the buffer */
static inline __attribute__((always_inline)) void put_bits(Bits* b, unsigned code, int len)
{
    b->acc = (b->acc << len) | code;
    b->n += len;
    if (b->n >= 32) {
        const uint32_t v = (uint32_t) (b->acc >> (b->n - 32));
        b->n -= 32;
        if (!(((~v) - 0x01010101u) & v & 0x80808080u) && b->end - b->p >= 4) {        /* no byte of v is 0xFF (no byte of ~v is zero) */
            b->p[0] = (uint8_t) (v >> 24); b->p[1] = (uint8_t) (v >> 16); b->p[2] = (uint8_t) (v >> 8); b->p[3] = (uint8_t) v;
            b->p += 4;
        } else {
            for (int s = 24; s >= 0; s -= 8) {
                const unsigned byte = (v >> s) & 0xFFu;
                put_byte(b, byte);
                if (byte == 0xFFu) put_byte(b, 0);
            }
        }
    }
}
/* what is left in the word at the end of the scan, the last byte padded with ones (F.1.2.3) */
static void flush_bits(Bits* b)
{
    if (b->n & 7) { const int pad = 8 - (b->n & 7); b->acc = (b->acc << pad) | ((1u << pad) - 1u); b->n += pad; }
    while (b->n >= 8) {
        const unsigned byte = (unsigned) (b->acc >> (b->n - 8)) & 0xFFu;
        put_byte(b, byte);
        if (byte == 0xFFu) put_byte(b, 0);
        b->n -= 8;
    }
}
static void put_marker(Bits* b, unsigned m, const uint8_t* body, int len)
{
    put_byte(b, 0xFF); put_byte(b, m);
    if (len >= 0) { put_byte(b, (unsigned) (len + 2) >> 8); put_byte(b, (unsigned) (len + 2) & 0xFF); for (int i = 0; i < len; i++) put_byte(b, body[i]); }
}

/* one block: quantise (round to nearest), DC difference + AC run lengths, Huffman (F.1.2) */
static inline __attribute__((always_inline)) void encode_block(Bits* b, float* blk, const float* rq, const Huff* dc, const Huff* ac, int* pred)
{
    int nat[64], q[64];
    fdct(blk);
    for (int i = 0; i < 64; i++) {                              /* (blk and rq both hold the transposed block) */
        const float v = blk[i] * rq[i];
        nat[i] = (int) (v + (v < 0.0f ? -0.5f : 0.5f));
    }
    for (int i = 0; i < 64; i++) q[i] = nat[ZIGZAG_T[i]];
    int diff = q[0] - *pred;
    *pred = q[0];
    {
        const int a = diff < 0 ? -diff : diff, nb = a ? 32 - __builtin_clz((unsigned) a) : 0;
        put_bits(b, dc->code[nb], dc->len[nb]);
        if (nb) put_bits(b, (unsigned) (diff < 0 ? diff - 1 : diff) & ((1u << nb) - 1u), nb);
    }
    int run = 0;
    for (int i = 1; i < 64; i++) {
        int v = q[i];
        if (v == 0) { run++; continue; }
        while (run > 15) { put_bits(b, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
        const int a = v < 0 ? -v : v;
        int nb = 32 - __builtin_clz((unsigned) a);
        if (nb > 10) { nb = 10; v = v < 0 ? -1023 : 1023; }     /* (cannot happen with 8-bit samples and q >= 1) */
        const int sym = (run << 4) | nb;
        put_bits(b, ac->code[sym], ac->len[sym]);
        put_bits(b, (unsigned) (v < 0 ? v - 1 : v) & ((1u << nb) - 1u), nb);
        run = 0;
    }
    if (run) put_bits(b, ac->code[0], ac->len[0]);              /* EOB */
}

size_t kbe_jpeg_bound(int w, int h)
{
    if (w <= 0 || h <= 0) return 0;
    const size_t mcus = (size_t) ((w + 15) / 16) * (size_t) ((h + 15) / 16);
    return 1024 + mcus * 6 * 64 * 4;                            /* headers + a generous 4 bytes per coefficient (16 code + 10 value bits, stuffed) */
}

KBE_HOT static int encode_one(const uint8_t* rgb, int w, int h, int stride, const Tables* t, uint8_t* out, size_t cap, size_t* size)
{
    Bits b = { out, out + cap, 0, 0, 0 };
    put_marker(&b, 0xD8, NULL, -1);                                                             /* SOI */
    { static const uint8_t jfif[14] = { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 }; put_marker(&b, 0xE0, jfif, 14); }
    for (int c = 0; c < 2; c++) {                                                               /* DQT, zigzag order */
        uint8_t body[65];
        body[0] = (uint8_t) c;
        for (int i = 0; i < 64; i++) body[1 + i] = t->q[c][ZIGZAG[i]];
        put_marker(&b, 0xDB, body, 65);
    }
    {                                                                                           /* SOF0: 8 bits, Y 2x2, Cb 1x1, Cr 1x1 */
        const uint8_t sof[15] = { 8, (uint8_t) (h >> 8), (uint8_t) h, (uint8_t) (w >> 8), (uint8_t) w, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1 };
        put_marker(&b, 0xC0, sof, 15);
    }
    {
        const struct { int id; const uint8_t* bits; const uint8_t* vals; int n; } dht[4] = {
            { 0x00, DC_LUMA_BITS, DC_VALS, 12 }, { 0x10, AC_LUMA_BITS, AC_LUMA_VALS, 162 }, { 0x01, DC_CHROMA_BITS, DC_VALS, 12 }, { 0x11, AC_CHROMA_BITS, AC_CHROMA_VALS, 162 } };
        for (int k = 0; k < 4; k++) {
            uint8_t body[1 + 16 + 162];
            body[0] = (uint8_t) dht[k].id;
            memcpy(body + 1, dht[k].bits, 16);
            memcpy(body + 17, dht[k].vals, (size_t) dht[k].n);
            put_marker(&b, 0xC4, body, 17 + dht[k].n);
        }
    }
    { static const uint8_t sos[10] = { 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0 }; put_marker(&b, 0xDA, sos, 10); }
    int pred[3] = { 0, 0, 0 };
    float Y[4][64], Cb[64], Cr[64];
    for (int my = 0; my < h; my += 16)
        for (int mx = 0; mx < w; mx += 16) {
            /* the MCU's 16 x 16 pixels (edge pixels repeated past the image) as three float planes, JFIF colour conversion, chroma averaged 2 x 2 */
            float r[16][16], g[16][16], bl[16][16], yy[16][16], cb[16][16], cr[16][16];
            const int inside = mx + 16 <= w && my + 16 <= h;
            for (int y = 0; y < 16; y++) {
                const uint8_t* row = rgb + (size_t) (my + y < h ? my + y : h - 1) * (size_t) stride;
                if (inside) {
                    const uint8_t* p = row + 3 * (size_t) mx;
                    for (int x = 0; x < 16; x++) { r[y][x] = p[3 * x]; g[y][x] = p[3 * x + 1]; bl[y][x] = p[3 * x + 2]; }
                } else
                    for (int x = 0; x < 16; x++) {
                        const uint8_t* p = row + 3 * (size_t) (mx + x < w ? mx + x : w - 1);
                        r[y][x] = p[0]; g[y][x] = p[1]; bl[y][x] = p[2];
                    }
            }
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) {
                    yy[y][x] = 0.299f * r[y][x] + 0.587f * g[y][x] + 0.114f * bl[y][x] - 128.0f;
                    cb[y][x] = -0.168735892f * r[y][x] - 0.331264108f * g[y][x] + 0.5f * bl[y][x];
                    cr[y][x] = 0.5f * r[y][x] - 0.418687589f * g[y][x] - 0.081312411f * bl[y][x];
                }
            for (int k = 0; k < 4; k++)
                for (int y = 0; y < 8; y++) memcpy(&Y[k][y * 8], &yy[(k >> 1) * 8 + y][(k & 1) * 8], 8 * sizeof(float));
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    Cb[y * 8 + x] = 0.25f * (cb[2 * y][2 * x] + cb[2 * y][2 * x + 1] + cb[2 * y + 1][2 * x] + cb[2 * y + 1][2 * x + 1]);
                    Cr[y * 8 + x] = 0.25f * (cr[2 * y][2 * x] + cr[2 * y][2 * x + 1] + cr[2 * y + 1][2 * x] + cr[2 * y + 1][2 * x + 1]);
                }
            for (int k = 0; k < 4; k++) encode_block(&b, Y[k], t->rq[0], &t->dc[0], &t->ac[0], &pred[0]);
            encode_block(&b, Cb, t->rq[1], &t->dc[1], &t->ac[1], &pred[1]);
            encode_block(&b, Cr, t->rq[1], &t->dc[1], &t->ac[1], &pred[2]);
        }
    flush_bits(&b);
    put_marker(&b, 0xD9, NULL, -1);                                                             /* EOI */
    if (b.overflow) return KBE_JPEG_E_SPACE;
    *size = (size_t) (b.p - out);
    return KBE_JPEG_OK;
}

int kbe_jpeg_encode(const uint8_t* rgb, int w, int h, int stride_bytes, int quality, uint8_t* out, size_t cap, size_t* size)
{
    if (!rgb || !out || !size || w <= 0 || h <= 0 || w > 65535 || h > 65535 || stride_bytes < 3 * w) return KBE_JPEG_E_INVALID;
    Tables t;
    tables_build(quality, &t);
    return encode_one(rgb, w, h, stride_bytes, &t, out, cap, size);
}

typedef struct {
    const uint8_t* const* rgb; uint8_t* const* outs; size_t* sizes; size_t cap;
    int n, w, h, stride; const Tables* t; int next; int status; pthread_mutex_t mu;
} Batch;

static void* batch_worker(void* arg)
{
    Batch* b = (Batch*) arg;
    for (;;) {
        pthread_mutex_lock(&b->mu);
        const int i = b->next < b->n ? b->next++ : -1;
        pthread_mutex_unlock(&b->mu);
        if (i < 0) return NULL;
        const int rc = encode_one(b->rgb[i], b->w, b->h, b->stride, b->t, b->outs[i], b->cap, &b->sizes[i]);
        if (rc != KBE_JPEG_OK) { pthread_mutex_lock(&b->mu); b->status = rc; pthread_mutex_unlock(&b->mu); }
    }
}

int kbe_jpeg_encode_batch(const uint8_t* const* rgb, int n, int w, int h, int stride_bytes, int quality, uint8_t* const* outs, size_t cap, size_t* sizes, int threads)
{
    if (n < 0 || (n > 0 && (!rgb || !outs || !sizes)) || w <= 0 || h <= 0 || w > 65535 || h > 65535 || stride_bytes < 3 * w) return KBE_JPEG_E_INVALID;
    for (int i = 0; i < n; i++) if (!rgb[i] || !outs[i]) return KBE_JPEG_E_INVALID;
    Tables t;
    tables_build(quality, &t);
    Batch b = { rgb, outs, sizes, cap, n, w, h, stride_bytes, &t, 0, KBE_JPEG_OK, PTHREAD_MUTEX_INITIALIZER };
    if (threads > n) threads = n;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    int started = 0;
    for (int k = 1; k < threads; k++)                           /* the caller's thread is one of them */
        if (pthread_create(&tid[started], NULL, batch_worker, &b) == 0) started++;
    batch_worker(&b);
    for (int k = 0; k < started; k++) pthread_join(tid[k], NULL);
    pthread_mutex_destroy(&b.mu);
    return b.status;
}
