/*
 * bb_oracle.c -- plain-C restatement of the reference decode algorithms.
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg; never by the product package.
 *
 * Parity status: pinned -- tests/test_oracle_c.py checks every function here
 * against the golden vectors captured from the real reference
 * (oracle/gen_golden.py) and against oracle/bb_oracle_np.py.
 *
 * Reference lines followed (paths relative to the reference tree):
 *   levels            base/encoding.py:14,46-56,131-144
 *   byte LUT + take   vdif/payload.py:25-103, mark5b/payload.py:27-94,
 *                     gsb/payload.py:24-42, dada/payload.py:13-14
 *   VDIF header bits  vdif/header.py:529-542
 *   frameset gather   vdif/frame.py:176-243,402-434; fill base/frame.py:191-199
 *   read loop         base/base.py:919-969, vdif/base.py:386-390
 *   Mark 5B           mark5b/header.py:60-68, mark5b/frame.py:62-70
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

enum { ORC_VDIF = 0, ORC_MARK5B = 1, ORC_INT = 2 };

static int lut_ready[3][9];
static float lut[3][9][256][8];

/* code -> level, as the NumPy expressions of the reference compute them */
static int code_levels(int coder, int bps, float *lev)
{
    const volatile float hi = 3.316505f;
    const volatile float s4 = 2.95f, s8 = 35.5f;
    int n;
    if (coder == ORC_VDIF) {
        if (bps == 1) { lev[0] = -1.f; lev[1] = 1.f; return 0; }
        if (bps == 2) { lev[0] = -hi; lev[1] = -1.f; lev[2] = 1.f; lev[3] = hi; return 0; }
        if (bps == 4) { for (n = 0; n < 16; ++n) { volatile float x = (float)n; x = x - 8.f; x = x / s4; lev[n] = x; } return 0; }
        if (bps == 8) { for (n = 0; n < 256; ++n) { volatile float x = (float)n; x = x - 127.5f; x = x / s8; lev[n] = x; } return 0; }
    } else if (coder == ORC_MARK5B) {
        if (bps == 1) { lev[0] = 1.f; lev[1] = -1.f; return 0; }
        if (bps == 2) { lev[0] = -hi; lev[1] = 1.f; lev[2] = -1.f; lev[3] = hi; return 0; }
    } else if (coder == ORC_INT) {
        if (bps == 4) { for (n = 0; n < 16; ++n) lev[n] = (float)(n < 8 ? n : n - 16); return 0; }
        if (bps == 8) { for (n = 0; n < 256; ++n) lev[n] = (float)(int8_t)(uint8_t)n; return 0; }
    }
    return -1;
}

static int ensure_lut(int coder, int bps)
{
    float lev[256];
    int b, i, per;
    if (coder < 0 || coder > 2 || bps < 1 || bps > 8) return -1;
    if (lut_ready[coder][bps]) return 0;
    if (code_levels(coder, bps, lev)) return -1;
    per = 8 / bps;
    for (b = 0; b < 256; ++b)
        for (i = 0; i < per; ++i)
            lut[coder][bps][b][i] = lev[(b >> (i * bps)) & ((1 << bps) - 1)];
    lut_ready[coder][bps] = 1;
    return 0;
}

/* lut.take(bytes): nbytes -> nbytes * 8/bps floats */
int orc_decode_flat(const uint8_t *raw, size_t nbytes, int coder, int bps, float *out)
{
    size_t i;
    int per = 8 / bps;
    if (ensure_lut(coder, bps)) return -1;
    for (i = 0; i < nbytes; ++i)
        memcpy(out + i * per, lut[coder][bps][raw[i]], sizeof(float) * per);
    return 0;
}

int orc_levels(int coder, int bps, float *lev) { return code_levels(coder, bps, lev); }

/*
 * Clean fixed-stride VDIF file -> (nsets*spf, nslot, chunk) float32, one
 * frame at a time like the reference's read loop.  thread_slot[1024] maps
 * thread_id -> output slot (or -1).  Returns the number of frames decoded,
 * or a negative value if a frame is out of place.
 */
long orc_vdif_read(const uint8_t *buf, size_t nbytes, int header_nbytes,
                   int frame_nbytes, int nthread_file, const int16_t *thread_slot,
                   int nslot, int coder, int bps, int chunk, int complex_data,
                   int frame_rate, float fill, float *out, size_t nsets)
{
    const int pn = frame_nbytes - header_nbytes;
    const size_t E = (size_t)pn * 8 / bps;
    const size_t R = E / chunk;
    const uint32_t *w0 = (const uint32_t *)buf;
    const int32_t s0 = (int32_t)(w0[0] & 0x3fffffff);
    const int32_t f0 = (int32_t)(w0[1] & 0xffffff);
    size_t set, r;
    long ndec = 0;
    int k, c, per = 8 / bps;
    float tmp[8];
    if (ensure_lut(coder, bps)) return -1;
    for (set = 0; set < nsets; ++set) {
        for (k = 0; k < nthread_file; ++k) {
            size_t o = (set * nthread_file + k) * (size_t)frame_nbytes;
            const uint32_t *w;
            const uint8_t *p;
            int slot;
            long idx;
            if (o + frame_nbytes > nbytes) return -2;
            w = (const uint32_t *)(buf + o);
            slot = thread_slot[(w[3] >> 16) & 0x3ff];
            if (slot < 0) continue;
            /* a set = the frames sharing the frame_nr of its first header, placed by
             * that header (VDIFFrameSet.fromfile, vdif/frame.py:201-243: the seconds
             * of the other threads are not looked at) */
            {
                const uint32_t *wl = (const uint32_t *)(buf + (set * nthread_file) * (size_t)frame_nbytes);
                if ((w[1] & 0xffffff) != (wl[1] & 0xffffff)) return -3;
                idx = (long)((int32_t)(wl[0] & 0x3fffffff) - s0) * frame_rate
                      + (int32_t)(wl[1] & 0xffffff) - f0;
            }
            if (idx != (long)set) return -3;
            p = buf + o + header_nbytes;
            if (w[0] >> 31) {                      /* invalid_data -> fill */
                for (r = 0; r < R; ++r) {
                    float *dst = out + ((set * R + r) * nslot + slot) * chunk;
                    for (c = 0; c < chunk; ++c)
                        dst[c] = (complex_data && (c & 1)) ? 0.f : fill;
                }
            } else {
                size_t e = 0, i;
                for (i = 0; i < (size_t)pn; ++i) {
                    int j;
                    memcpy(tmp, lut[coder][bps][p[i]], sizeof(float) * per);
                    for (j = 0; j < per; ++j, ++e) {
                        size_t row = e / chunk, within = e % chunk;
                        out[((set * R + row) * nslot + slot) * chunk + within] = tmp[j];
                    }
                }
            }
            ++ndec;
        }
    }
    return ndec;
}

/* Clean Mark 5B file -> (nframes*spf, nchan) float32 */
long orc_mark5b_read(const uint8_t *buf, size_t nbytes, int bps, float fill,
                     float *out, size_t nframes)
{
    const size_t E = 10000u * 8 / bps;
    size_t f, i;
    int per = 8 / bps;
    if (ensure_lut(ORC_MARK5B, bps)) return -1;
    for (f = 0; f < nframes; ++f) {
        const uint8_t *fr = buf + f * 10016u;
        const uint32_t *w = (const uint32_t *)fr;
        int all_fill = 1;
        if ((f + 1) * 10016u > nbytes) return -2;
        if (w[0] != 0xABADDEEDu) return -3;
        for (i = 0; i < 2500; ++i)
            if (w[4 + i] != 0x11223344u) { all_fill = 0; break; }
        if (all_fill) {
            for (i = 0; i < E; ++i) out[f * E + i] = fill;
        } else {
            for (i = 0; i < 10000; ++i)
                memcpy(out + f * E + i * per, lut[ORC_MARK5B][bps][fr[16 + i]],
                       sizeof(float) * per);
        }
    }
    return (long)nframes;
}
