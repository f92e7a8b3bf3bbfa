/*
 * san_check.c -- the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer
 * (SURVEY.md section 5, row "sanitizers": GPU ASan is not available on this
 * pool, the CPU build is what can be checked).  Test infrastructure, like the
 * rest of oracle/.
 *
 * Every buffer is malloc'ed at its exact size, so an over-read or over-write of
 * a single byte aborts the program; UBSan aborts on misaligned access, shifts
 * out of range and signed overflow (-fno-sanitize-recover).  The decoders are
 * run on random input at edge sizes and their outputs compared with values
 * computed here from the level tables, bit for bit.
 *
 * Build + run:  make -C oracle san     (prints "san_check ok")
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_decode_flat(const uint8_t *raw, size_t nbytes, int coder, int bps, float *out);
int orc_levels(int coder, int bps, float *lev);
long orc_vdif_read(const uint8_t *buf, size_t nbytes, int header_nbytes, int frame_nbytes,
                   int nthread_file, const int16_t *thread_slot, int nslot, int coder, int bps,
                   int chunk, int complex_data, int frame_rate, float fill, float *out, size_t nsets);
long orc_mark5b_read(const uint8_t *buf, size_t nbytes, int bps, float fill, float *out, size_t nframes);

static uint32_t rng_state = 12345u;
static uint32_t rnd(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return rng_state;
}

static int fail(const char *what, long a, long b)
{
    fprintf(stderr, "san_check: %s (%ld, %ld)\n", what, a, b);
    return 1;
}

static int same_bits(float a, float b) { return memcmp(&a, &b, 4) == 0; }

int main(void)
{
    static const int combos[][2] = {{0, 1}, {0, 2}, {0, 4}, {0, 8}, {1, 1}, {1, 2}, {2, 4}, {2, 8}};
    static const size_t sizes[] = {0, 1, 3, 4, 5, 255, 256, 1000, 8000, 10000, 65537};
    size_t ci, si, i;
    /* 1. flat decode, exact-size buffers */
    for (ci = 0; ci < sizeof(combos) / sizeof(combos[0]); ++ci) {
        const int coder = combos[ci][0], bps = combos[ci][1], per = 8 / bps;
        float lev[256];
        if (orc_levels(coder, bps, lev)) return fail("levels", coder, bps);
        for (si = 0; si < sizeof(sizes) / sizeof(sizes[0]); ++si) {
            const size_t n = sizes[si];
            uint8_t *raw = (uint8_t *)malloc(n ? n : 1);
            float *out = (float *)malloc((n ? n : 1) * per * sizeof(float));
            for (i = 0; i < n; ++i) raw[i] = (uint8_t)rnd();
            if (orc_decode_flat(raw, n, coder, bps, out)) return fail("decode_flat rc", coder, bps);
            for (i = 0; i < n * per; ++i) {
                const unsigned code = (raw[i / per] >> (bps * (i % per))) & ((1u << bps) - 1u);
                if (!same_bits(out[i], lev[code])) return fail("decode_flat value", (long)n, (long)i);
            }
            free(raw); free(out);
        }
    }
    /* 2. VDIF read loop: 3 threads on disk in shuffled order, 2 selected, an
     *    invalid frame, complex 2-bit data with 2 channels */
    {
        const int hn = 32, pn = 64, fn = hn + pn, nth = 3, nsets = 5, nslot = 2, chunk = 4, bps = 2;
        const int order[3] = {2, 0, 1};
        const size_t nbytes = (size_t)fn * nth * nsets;
        const size_t E = (size_t)pn * 8 / bps, R = E / chunk;
        uint8_t *buf = (uint8_t *)malloc(nbytes);
        float *out = (float *)malloc(sizeof(float) * nsets * R * nslot * chunk);
        int16_t slot[1024];
        float lev[4];
        long rc;
        int s, k;
        for (i = 0; i < 1024; ++i) slot[i] = -1;
        slot[0] = 0; slot[2] = 1;                       /* thread 1 is not selected */
        orc_levels(0, 2, lev);
        for (i = 0; i < nbytes; ++i) buf[i] = (uint8_t)rnd();
        for (s = 0; s < nsets; ++s)
            for (k = 0; k < nth; ++k) {
                uint32_t w[8] = {0};
                w[0] = 100u + (uint32_t)(s / 2);        /* seconds; frame rate 2 */
                w[1] = (uint32_t)(s % 2);
                w[2] = (uint32_t)(fn / 8);
                w[3] = (1u << 31) | (1u << 26) | ((uint32_t)order[k] << 16);
                if (s == 3 && order[k] == 2) w[0] |= 1u << 31;     /* invalid */
                memcpy(buf + ((size_t)s * nth + k) * fn, w, 32);
            }
        rc = orc_vdif_read(buf, nbytes, hn, fn, nth, slot, nslot, 0, bps, chunk, 1, 2, -5.f, out, nsets);
        if (rc != nsets * 2) return fail("vdif_read frames", rc, nsets * 2);
        for (s = 0; s < nsets; ++s)
            for (k = 0; k < nth; ++k) {
                const int sl = slot[order[k]];
                const uint8_t *p = buf + ((size_t)s * nth + k) * fn + hn;
                size_t e;
                if (sl < 0) continue;
                for (e = 0; e < E; ++e) {
                    const float got = out[((s * R + e / chunk) * nslot + sl) * chunk + e % chunk];
                    float want = lev[(p[e / 4] >> (2 * (e % 4))) & 3];
                    if (s == 3 && order[k] == 2) want = (e & 1) ? 0.f : -5.f;
                    if (!same_bits(got, want)) return fail("vdif_read value", s, (long)e);
                }
            }
        /* a truncated file is refused, not read past its end */
        rc = orc_vdif_read(buf, nbytes - 1, hn, fn, nth, slot, nslot, 0, bps, chunk, 1, 2, 0.f, out, nsets);
        if (rc != -2) return fail("vdif_read truncated", rc, -2);
        free(buf); free(out);
    }
    /* 3. Mark 5B read loop with a fill-pattern frame */
    {
        const size_t nfr = 3, fn = 10016, E = 40000;
        uint8_t *buf = (uint8_t *)malloc(nfr * fn);
        float *out = (float *)malloc(sizeof(float) * nfr * E);
        float lev[4];
        size_t f;
        long rc;
        orc_levels(1, 2, lev);
        for (i = 0; i < nfr * fn; ++i) buf[i] = (uint8_t)rnd();
        for (f = 0; f < nfr; ++f) {
            const uint32_t sync = 0xABADDEEDu;
            memcpy(buf + f * fn, &sync, 4);
        }
        for (i = 0; i < 2500; ++i) {
            const uint32_t fillw = 0x11223344u;
            memcpy(buf + fn + 16 + 4 * i, &fillw, 4);
        }
        rc = orc_mark5b_read(buf, nfr * fn, 2, 7.f, out, nfr);
        if (rc != (long)nfr) return fail("mark5b_read rc", rc, (long)nfr);
        for (f = 0; f < nfr; ++f)
            for (i = 0; i < E; ++i) {
                const float want = f == 1 ? 7.f : lev[(buf[f * fn + 16 + i / 4] >> (2 * (i % 4))) & 3];
                if (!same_bits(out[f * E + i], want)) return fail("mark5b_read value", (long)f, (long)i);
            }
        if (orc_mark5b_read(buf, nfr * fn - 1, 2, 7.f, out, nfr) != -2) return fail("mark5b truncated", 0, 0);
        free(buf); free(out);
    }
    printf("san_check ok\n");
    return 0;
}
