// Mark 4: embedded-header scan and track-demultiplexing decode.
//
// A Mark 4 frame is 20000 "stream words" of ntrack bits (ntrack = 16/32/64);
// bit k of every word belongs to tape track k.  The first 160 words carry the
// per-track headers (bit k of word j = header bit j of track k, MSB first:
// mark4/header.py:47-64), the rest the samples: each word holds `fanout`
// consecutive samples of `nchan` channels as (sign, magnitude) bit pairs on
// track positions given by the track-assignment tables (mark4/header.py:306-
// 328) -- i.e. the output of one word is fanout*nchan = ntrack/2 contiguous
// float32 values.
//
// Replaces (reference, path:line): Mark4Header.fromfile + stream2words
// (mark4/header.py:425-454,47-64), the sync search pattern
// (mark4/header.py:345-373), frame validity from the error flags
// (mark4/frame.py:78-87), the five decoders reorder32/64/64_Ft + lut2bit{1,3}
// + transposes (mark4/payload.py:48-69,122-288), and the header-overwrite
// fill of Mark4Frame.__getitem__ (mark4/frame.py:185-189,248-258).
#pragma once
#include "bb_common.h"
#include "k_scan.h"

template <int NTRACK> struct bb_m4_word;
template <> struct bb_m4_word<16> { typedef uint16_t type; };
template <> struct bb_m4_word<32> { typedef uint32_t type; };
template <> struct bb_m4_word<64> { typedef uint64_t type; };

__device__ __forceinline__ uint32_t bb_bcd(uint32_t v, int ndigit, bool *ok)
{
    uint32_t r = 0, m = 1;
    for (int i = 0; i < ndigit; ++i) {
        const uint32_t d = (v >> (4 * i)) & 0xf;
        if (d > 9) *ok = false;
        r += d * m;
        m *= 10;
    }
    return r;
}

// One wave per frame.  Lane i looks at stream words 32+i (flags, sync) and
// 96+i (time code); three ballots give validity, sync and the 64 time-code
// bits of track 0.
// Stream word at any byte position (frames found by bb_mark4_locate need not
// be word aligned): assembled from bytes, header reads only.
template <int NTRACK>
__device__ __forceinline__ typename bb_m4_word<NTRACK>::type
bb_m4_load_any(const uint8_t *buf, uint64_t pos)
{
    typedef typename bb_m4_word<NTRACK>::type word_t;
    if ((pos & (sizeof(word_t) - 1)) == 0)
        return *reinterpret_cast<const word_t *>(buf + pos);
    word_t w = 0;
#pragma unroll
    for (int k = 0; k < (int)sizeof(word_t); ++k) w |= (word_t)buf[pos + k] << (8 * k);
    return w;
}

// Byte-granular Mark 4 frame search: position p holds a frame
// when stream word 63 is all zero and words 64..95 are all ones (the sync
// pattern of every track plus the zero bit before it: mark4/header.py:345-373),
// the whole frame fits, and -- if its pattern still fits in the buffer -- the
// frame one later shows the same: locate_frames with check=1
// (mark4/base.py:110-166, base/base.py:181-335).  One lane per position; the
// first byte (0xff) rejects nearly all of them.
template <int NTRACK>
__device__ __forceinline__ bool bb_m4_sync_at(const uint8_t *buf, uint64_t pos)
{
    constexpr int ISZ = NTRACK / 8;
    const uint8_t *q = buf + pos + 63 * ISZ;
    if (q[ISZ] != 0xffu) return false;
#pragma unroll
    for (int k = 0; k < ISZ; ++k) if (q[k] != 0) return false;
    for (int k = 1; k < 32 * ISZ; ++k) if (q[ISZ + k] != 0xffu) return false;
    return true;
}

// The sweep (bb_locate_sweep, k_scan.h) probes for the END of the zero word:
// a zero byte followed by three 0xff bytes, i.e. the dword 0xffffff00 at
// z = pos + 64 * ISZ - 1.
template <int NTRACK>
__global__ __launch_bounds__(BB_BLOCK)
void k_mark4_locate(const uint8_t *buf, uint64_t nbytes, int64_t *out, uint64_t cap,
                    unsigned long long *count)
{
    constexpr uint64_t ISZ = NTRACK / 8;
    constexpr uint64_t FN = (uint64_t)NTRACK * 2500, PAT_END = 96 * ISZ, ZOFF = 64 * ISZ - 1;
    const uint64_t q_end = nbytes - FN + ZOFF + 1;
    bb_locate_sweep(buf, nbytes, q_end, out, cap, count,
        [&](uint32_t v) { return v ^ 0xffffff00u; },
        [&](uint64_t z) -> int64_t {
            if (z < ZOFF) return -1;
            const uint64_t pos = z - ZOFF;
            if (pos + FN > nbytes || !bb_m4_sync_at<NTRACK>(buf, pos)) return -1;
            if (pos + FN + PAT_END <= nbytes && !bb_m4_sync_at<NTRACK>(buf, pos + FN)) return -1;
            return (int64_t)pos;
        });
}

template <int NTRACK>
__global__ __launch_bounds__(BB_BLOCK)
void k_mark4_scan(const uint8_t *buf, uint64_t nbytes, bb_mark4_scan_params p,
                  const int64_t *offsets, bb_frame_rec *recs, uint64_t nframes)
{
    typedef typename bb_m4_word<NTRACK>::type word_t;
    const uint64_t frame = (uint64_t)blockIdx.x * BB_WAVES_PER_BLOCK + bb_wave();
    if (frame >= nframes) return;                       // wave-uniform
    const int lane = bb_lane();
    const uint64_t frame_nbytes = (uint64_t)NTRACK * 2500;
    // fixed stride, or explicit (possibly odd) offsets from bb_mark4_locate
    const uint64_t off = offsets ? (uint64_t)offsets[frame] : p.first_offset + frame * frame_nbytes;
    const bool whole = off + frame_nbytes <= nbytes;
    word_t a = 0, b = 0;
    if (whole) {
        a = bb_m4_load_any<NTRACK>(buf, off + (uint64_t)(32 + lane) * sizeof(word_t));
        b = bb_m4_load_any<NTRACK>(buf, off + (uint64_t)(96 + lane) * sizeof(word_t));
    }
    const word_t ones = (word_t)~(word_t)0;
    // error flags: header word 1 bits 15..12 -> stream words 48..51
    const bool err = lane >= 16 && lane < 20 && a != 0;
    // sync: stream word 63 all zero, words 64..95 all ones
    const bool syn = lane < 31 ? true : (lane == 31 ? a == 0 : a == ones);
    const unsigned long long errs = __ballot(err);
    const unsigned long long syns = __ballot(syn);
    const unsigned long long tb = __ballot((b & 1) != 0);
    if (lane == 0) {
        const uint32_t w3 = __brev((uint32_t)(tb & 0xffffffffull));
        const uint32_t w4 = __brev((uint32_t)(tb >> 32));
        bool ok = whole && syns == ~0ull;
        const uint32_t uyear = (w3 >> 28) & 0xf;
        const uint32_t day = bb_bcd((w3 >> 16) & 0xfff, 3, &ok);
        const uint32_t hour = bb_bcd((w3 >> 8) & 0xff, 2, &ok);
        const uint32_t minute = bb_bcd(w3 & 0xff, 2, &ok);
        const uint32_t sec = bb_bcd((w4 >> 24) & 0xff, 2, &ok);
        const uint32_t ms = bb_bcd((w4 >> 12) & 0xfff, 3, &ok);
        // last ms digit d encodes d*1.25 ms (mark4/header.py:198-214)
        int64_t q = (((int64_t)(day * 24 + hour) * 60 + minute) * 60 + sec) * 4000
                    + 4 * ms + ms % 5;
        const int y = p.ref_year;
        if (uyear == (uint32_t)((y + 1) % 10) && uyear != (uint32_t)(y % 10)) {
            const int leap = (y % 4 == 0 && (y % 100 != 0 || y % 400 == 0)) ? 1 : 0;
            q += (int64_t)(365 + leap) * 86400 * 4000;
        } else if (uyear != (uint32_t)(y % 10)) {
            ok = false;
        }
        int64_t tidx;
        if (p.by_position) {
            ok = whole;                                 // (verify=False: by where it lies, nothing checked)
            tidx = (int64_t)frame;
        } else if (p.frame_qms > 0) {
            const int64_t dq = q - p.ref_qms;
            tidx = (dq >= 0 ? dq + p.frame_qms / 2 : dq - p.frame_qms / 2) / p.frame_qms;
            if (tidx * p.frame_qms != dq) ok = false;   // not on the frame grid
        } else {
            tidx = (int64_t)frame;
        }
        if (tidx > 0x7fffffffll) tidx = 0x7fffffffll;
        if (tidx < -0x7fffffffll) tidx = -0x7fffffffll;
        bb_frame_rec r;
        r.payload_offset = (int64_t)off;                // frame start: decode skips the header words
        r.time_index = (int32_t)tidx;
        r.thread_id = 0;
        r.flags = (uint16_t)((ok ? BB_FRAME_OK : 0u) | (errs ? BB_FRAME_INVALID : 0u));
        *reinterpret_cast<bb_u4 *>(&recs[frame]) = *reinterpret_cast<const bb_u4 *>(&r);
    }
}

// Longitudinal (along-track) check of the frame headers: the 160 header bits of
// every track end in a CRC-12 (x^12 + x^11 + x^3 + x^2 + x + 1, 0x180f:
// mark4/header.py:34-44; CRCStack, base/utils.py:200-248), so the 160-bit
// stream of a sound track divides by the polynomial.  The reference computes
// this for all tracks at once on stream words (`crc12.check(stream)`,
// mark4/tests/test_mark4.py:57-58) but never applies it while reading; here it
// is an extra that reports, per frame, the set of tracks whose header fails --
// it never alters decoded samples.  Bit-sliced like
// the reference: twelve NTRACK-bit registers hold one remainder bit of every
// track each; one thread walks the 160 stream words of one frame.
template <int NTRACK>
__global__ __launch_bounds__(BB_BLOCK)
void k_mark4_header_crc(const uint8_t *buf, uint64_t nbytes, const int64_t *offsets,
                        int64_t first_offset, uint64_t nframes, uint64_t *bad_tracks)
{
    typedef typename bb_m4_word<NTRACK>::type word_t;
    const uint64_t f = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    if (f >= nframes) return;
    const uint64_t fn = (uint64_t)NTRACK * 2500;
    const uint64_t off = offsets ? (uint64_t)offsets[f] : (uint64_t)first_offset + f * fn;
    if (off + 160 * sizeof(word_t) > nbytes) { bad_tracks[f] = ~0ull >> (64 - NTRACK); return; }
    word_t r[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) r[k] = 0;
    for (int j = 0; j < 160; ++j) {
        const word_t in = bb_m4_load_any<NTRACK>(buf, off + (uint64_t)j * sizeof(word_t));
        const word_t fb = (word_t)(r[11] ^ in);
        r[11] = (word_t)(r[10] ^ fb);               // x^11
#pragma unroll
        for (int k = 10; k >= 4; --k) r[k] = r[k - 1];
        r[3] = (word_t)(r[2] ^ fb);                 // x^3
        r[2] = (word_t)(r[1] ^ fb);                 // x^2
        r[1] = (word_t)(r[0] ^ fb);                 // x
        r[0] = fb;                                  // 1
    }
    word_t bad = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) bad |= r[k];
    bad_tracks[f] = (uint64_t)bad;
}

struct bb_m4_args {
    const uint8_t *buf;
    const int64_t *src;
    float         *out;
    uint64_t nframes;
    uint64_t nwords;        // stream words per frame (20000) or per payload
    uint64_t fill_words;    // leading words that decode to fill (160 / 0)
    uint64_t nseg;
    uint32_t seg_tiles, tpw;
    int64_t  src0, src_stride;
    uint32_t sign_bit[8];   // 32 x uint8: output j -> bit position of its sign
    uint32_t mag_bit[8];    //                       ... of its magnitude
    float    fill, hi;
    uint64_t src_lim;       // offsets outside [0, src_lim) decode as fill (bb_src_ok)
    bb_perm_t perm;         // work order (bb_common.h)
};

#define BB_M4_SEG_TILES 32
#define BB_M4_TPW (BB_M4_SEG_TILES / BB_WAVES_PER_BLOCK)

// Lane l of a wave owns outputs [4*(l % LPW), +4) of word (l / LPW) of the
// current pass, LPW = NTRACK/8 lanes per word, so a store instruction covers
// 64/LPW consecutive words = 1 KiB of contiguous output.  Words are loaded
// once per 64-word tile (one coalesced load) and handed around with shuffles.
// Persistent and software pipelined like k_decode_flat_pipe: the loads of the
// next work item (8 tiles per wave) are in flight while the current one is
// being stored.
template <int NTRACK, bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_mark4(bb_m4_args a)
{
    typedef typename bb_m4_word<NTRACK>::type word_t;
    constexpr int LPW = NTRACK / 8;         // lanes per word
    constexpr int WPP = 64 / LPW;           // words per store pass
    constexpr int OPW = NTRACK / 2;         // outputs per word
    constexpr int TPW = BB_M4_TPW;
    const int lane = bb_lane();
    const int wave = bb_wave();
    const int sub = lane % LPW;
    const int wsel = lane / LPW;
    // per-lane bit positions of its four outputs (uniform selects, once)
    uint32_t spack = 0, mpack = 0;
#pragma unroll
    for (int k = 0; k < LPW; ++k)
        if (sub == k) { spack = a.sign_bit[k]; mpack = a.mag_bit[k]; }
    const float hi = a.hi;
    const bb_f4 fillv = {a.fill, a.fill, a.fill, a.fill};
    const uint64_t E = a.nwords * OPW;
    const uint64_t nwork = a.nframes * a.nseg;

    word_t cur[TPW], nxt[TPW];
    bool cur_valid = false, nxt_valid = false;

    auto issue = [&](uint64_t step, word_t (&w)[TPW], bool &valid) {
        const uint64_t work = bb_perm(a.perm, step);
        uint64_t f, seg;
        if (a.nseg == 1) { f = work; seg = 0; }
        else { f = work / a.nseg; seg = work - f * a.nseg; }
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        valid = bb_src_ok(so, a.src_lim);
        const word_t *in = reinterpret_cast<const word_t *>(a.buf + (valid ? so : 0));
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t w_end = (seg + 1) * a.seg_tiles * 64 < a.nwords
                               ? (seg + 1) * a.seg_tiles * 64 : a.nwords;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t wi = (tile0 + u) * 64 + lane;
            w[u] = (valid && u < (int)a.tpw && wi < w_end && wi >= a.fill_words) ? in[wi] : (word_t)0;
        }
    };

    uint64_t work = blockIdx.x;
    if (work < nwork) issue(work, cur, cur_valid);
    for (; work < nwork; work += gridDim.x) {
        const uint64_t next = work + gridDim.x;
        if (next < nwork) issue(next, nxt, nxt_valid);
        const uint64_t pwork = bb_perm(a.perm, work);
        uint64_t f, seg;
        if (a.nseg == 1) { f = pwork; seg = 0; }
        else { f = pwork / a.nseg; seg = pwork - f * a.nseg; }
        float *obase = a.out + f * E;
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t w_end = (seg + 1) * a.seg_tiles * 64 < a.nwords
                               ? (seg + 1) * a.seg_tiles * 64 : a.nwords;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t tile = tile0 + u;
            const bool live = u < (int)a.tpw;           // wave-uniform; no break: keep cur[] in registers
            const word_t w = cur[u];
#pragma unroll
            for (int p = 0; p < LPW; ++p) {
                const int srcl = p * WPP + wsel;
                uint64_t x;
                if (NTRACK == 64) {
                    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)((uint64_t)w & 0xffffffffull), srcl);
                    const uint32_t hi32 = (uint32_t)__shfl((int)(uint32_t)((uint64_t)w >> 32), srcl);
                    x = ((uint64_t)hi32 << 32) | lo;
                } else {
                    x = (uint32_t)__shfl((int)(uint32_t)w, srcl);
                }
                const uint64_t widx = tile * 64 + srcl;
                if (!live || widx >= w_end) continue;
                bb_f4 v;
                if (!cur_valid || widx < a.fill_words) {
                    v = fillv;
                } else {
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t sb = (spack >> (8 * k)) & 0xff;
                        const uint32_t mb = (mpack >> (8 * k)) & 0xff;
                        const bool s = (x >> sb) & 1;
                        const bool m = (x >> mb) & 1;
                        // sign set = positive; magnitude set = high level
                        // (mark4/payload.py:93-115: index 2*s + m into
                        // {-Hi, -1, +1, +Hi})
                        r[k] = (s == m) ? (s ? hi : -hi) : (s ? 1.0f : -1.0f);
                    }
                    v = bb_f4{r[0], r[1], r[2], r[3]};
                }
                bb_store4<NT>(obase + widx * OPW + 4 * sub, v);
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) cur[u] = nxt[u];
        cur_valid = nxt_valid;
    }
}

#if BB_EXP
// (experiment build; measured and not kept: profiles/r04s_exp_m4lds.log)
// The same decode for 64-bit stream words (64 tracks, and the narrower modes as
// super-words) with the words staged in LDS by direct-to-LDS loads
// (global_load_lds_dwordx4; round 4, as in k_lds.h): a wave brings its (up to)
// 8 tiles x 512 bytes in with four 1 KiB load instructions, no VGPR round trip
// and no shuffles -- a store pass reads its word from LDS (ds_read_b64; the
// eight lanes of a word read the same address: a broadcast).  Words of the fill
// prefix and of invalid frames are not loaded.  Units at addresses that are not
// multiples of 16 bytes keep every load aligned (the image starts at the
// 16-byte boundary below the first word); pieces that are not entirely inside
// the wave's words are loaded dword by dword, units at odd addresses byte by byte.
template <bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_mark4_lds(bb_m4_args a)
{
    constexpr int LPW = 8, WPP = 8, OPW = 32, TPW = BB_M4_TPW;
    constexpr int NPIECE = TPW * 32 + 1;                    // 16-byte pieces a wave stages at most
    __shared__ bb_u4 s_stage[BB_WAVES_PER_BLOCK][NPIECE];
    const int lane = bb_lane();
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    const int sub = lane % LPW;
    const int wsel = lane / LPW;
    uint32_t spack = 0, mpack = 0;
#pragma unroll
    for (int k = 0; k < LPW; ++k)
        if (sub == k) { spack = a.sign_bit[k]; mpack = a.mag_bit[k]; }
    const float hi = a.hi;
    const bb_f4 fillv = {a.fill, a.fill, a.fill, a.fill};
    const uint64_t E = a.nwords * OPW;
    const uint64_t nwork = a.nframes * a.nseg;
    const uint8_t *stage8 = reinterpret_cast<const uint8_t *>(&s_stage[wave][0]);
    uint32_t *stage32 = reinterpret_cast<uint32_t *>(&s_stage[wave][0]);

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        uint64_t f, seg;
        if (a.nseg == 1) { f = work; seg = 0; }
        else { f = work / a.nseg; seg = work - f * a.nseg; }
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t w_end = (seg + 1) * a.seg_tiles * 64 < a.nwords ? (seg + 1) * a.seg_tiles * 64 : a.nwords;
        uint64_t w0 = tile0 * 64;                               // first word of this wave
        uint64_t w1 = w0 + (uint64_t)a.tpw * 64 < w_end ? w0 + (uint64_t)a.tpw * 64 : w_end;
        // words to load: [wl, w1), without the fill prefix
        const uint64_t wl = w0 > a.fill_words ? w0 : a.fill_words;
        uint32_t sh = 0;                                        // byte of word w0 in the staged image
        if (valid && wl < w1) {
            const uint8_t *p0 = a.buf + (uint64_t)so + w0 * 8;  // word w0 (may lie in the fill prefix: not read)
            if (reinterpret_cast<uintptr_t>(p0) & 3) {
                uint8_t *st8 = reinterpret_cast<uint8_t *>(&s_stage[wave][0]);
                const uint32_t lo8 = (uint32_t)(wl - w0) * 8, hi8 = (uint32_t)(w1 - w0) * 8;
#pragma nounroll
                for (uint32_t i = lo8 + (uint32_t)lane; i < hi8; i += BB_WAVE) st8[i] = p0[i];
            } else {
                sh = (uint32_t)(reinterpret_cast<uintptr_t>(p0) & 15);
                const uint8_t *base = p0 - sh;                  // 16-byte aligned
                const uint32_t lo = sh + (uint32_t)(wl - w0) * 8, hiB = sh + (uint32_t)(w1 - w0) * 8;
#pragma unroll
                for (int k = 0; k < (NPIECE + BB_WAVE - 1) / BB_WAVE; ++k) {
                    const uint32_t piece = (uint32_t)k * BB_WAVE + (uint32_t)lane;
                    const uint32_t b0 = piece * 16;
                    if (piece >= (uint32_t)NPIECE || b0 + 16 <= lo || b0 >= hiB) continue;
                    if (b0 >= lo && b0 + 16 <= hiB) {
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(base + b0),
                            (__attribute__((address_space(3))) void *)(&s_stage[wave][k * BB_WAVE]), 16, 0, 0);
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t q = b0 + 4 * d;
                            if (q >= lo && q + 4 <= hiB) stage32[piece * 4 + d] = *reinterpret_cast<const uint32_t *>(base + q);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float *obase = a.out + f * E;
        // (tiles one after the other, the eight passes of a tile unrolled: all 64 unrolled
        // took 251 VGPRs = one wave per SIMD)
#pragma nounroll
        for (int u = 0; u < (int)a.tpw; ++u) {
#pragma unroll
            for (int p = 0; p < LPW; ++p) {
                const uint32_t wt = (uint32_t)u * 64 + (uint32_t)(p * WPP + wsel);     // word of this wave
                const uint64_t widx = w0 + wt;
                if (widx >= w1) continue;
                bb_f4 v = fillv;
                if (valid && widx >= a.fill_words) {
                    const uint32_t off = sh + wt * 8;
                    const uint64_t x = (uint64_t)*reinterpret_cast<const uint32_t *>(stage8 + off)
                                     | ((uint64_t)*reinterpret_cast<const uint32_t *>(stage8 + off + 4) << 32);
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t sb = (spack >> (8 * k)) & 0xff;
                        const uint32_t mb = (mpack >> (8 * k)) & 0xff;
                        const bool s = (x >> sb) & 1;
                        const bool m = (x >> mb) & 1;
                        r[k] = (s == m) ? (s ? hi : -hi) : (s ? 1.0f : -1.0f);
                    }
                    v = bb_f4{r[0], r[1], r[2], r[3]};
                }
                bb_store4<NT>(obase + widx * OPW + 4 * sub, v);
            }
        }
        __builtin_amdgcn_wave_barrier();          // the next step overwrites the staging area
    }
}
#endif

// Channel selection folded into the track demultiplexing (a reader `subset`
// that picks channels: the reference decodes whole frames and indexes
// afterwards, base/base.py:706-717 with 957-969).  The bit maps are data, so
// a selection is just a SHORTER map: `nout` = fanout x (selected channels)
// outputs per stream word instead of NTRACK/2, output j of a word taken from
// (sign_bit[j], mag_bit[j]).  A wave parks its 64-word tile in LDS; the tile's
// 64 * nout floats are then written as consecutive float4 (lane -> float4 ->
// up to four (word, output) pairs, each an LDS read and two bit tests), so
// HBM sees contiguous 16-byte stores whatever `nout` is.  When a unit's float
// count is not a multiple of four (or the output is only 4-byte aligned) the
// same walk is done float by float.
template <int NTRACK, bool NT, bool V4>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_mark4_select(bb_m4_args a, uint32_t nout)
{
    typedef typename bb_m4_word<NTRACK>::type word_t;
    __shared__ word_t s_w[BB_WAVES_PER_BLOCK][64];
    __shared__ uint8_t s_sb[32], s_mb[32];
    const int lane = bb_lane();
    const int wave = bb_wave();
    if (threadIdx.x < 32) {
        s_sb[threadIdx.x] = (uint8_t)(a.sign_bit[threadIdx.x / 4] >> (8 * (threadIdx.x % 4)));
        s_mb[threadIdx.x] = (uint8_t)(a.mag_bit[threadIdx.x / 4] >> (8 * (threadIdx.x % 4)));
    }
    const float hi = a.hi, fill = a.fill;
    const uint64_t E = a.nwords * nout;
    const uint64_t nwork = a.nframes * a.nseg;
    auto level = [&](word_t x, uint32_t j) -> float {
        const bool s = (x >> s_sb[j]) & 1;
        const bool m = (x >> s_mb[j]) & 1;
        return (s == m) ? (s ? hi : -hi) : (s ? 1.0f : -1.0f);
    };
    for (uint64_t work = blockIdx.x; work < nwork; work += gridDim.x) {
        const uint64_t pwork = bb_perm(a.perm, work);
        uint64_t f, seg;
        if (a.nseg == 1) { f = pwork; seg = 0; }
        else { f = pwork / a.nseg; seg = pwork - f * a.nseg; }
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        float *obase = a.out + f * E;
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t w_end = (seg + 1) * a.seg_tiles * 64 < a.nwords
                               ? (seg + 1) * a.seg_tiles * 64 : a.nwords;
        // (every wave makes all a.tpw rounds: the barriers are workgroup wide)
        for (uint32_t u = 0; u < a.tpw; ++u) {
            const uint64_t wbase = (tile0 + u) * 64;
            const uint64_t wi = wbase + lane;
            word_t w = 0;
            if (valid && wi < w_end && wi >= a.fill_words)
                w = bb_m4_load_any<NTRACK>(a.buf, (uint64_t)so + wi * sizeof(word_t));
            __syncthreads();                    // the previous tile has been read
            s_w[wave][lane] = w;
            __syncthreads();
            if (wbase >= w_end) continue;
            const uint32_t nw = w_end - wbase < 64 ? (uint32_t)(w_end - wbase) : 64u;
            const uint32_t nfl = nw * nout;
            float *o = obase + wbase * nout;
            if (V4) {
                for (uint32_t q = 4 * lane; q < nfl; q += 256) {
                    uint32_t wl = q / nout, j = q - wl * nout;
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        r[k] = (!valid || wbase + wl < a.fill_words) ? fill : level(s_w[wave][wl], j);
                        if (++j == nout) { j = 0; ++wl; }
                    }
                    bb_store4<NT>(o + q, bb_f4{r[0], r[1], r[2], r[3]});
                }
            } else {
                for (uint32_t q = lane; q < nfl; q += 64) {
                    const uint32_t wl = q / nout, j = q - wl * nout;
                    bb_store1<NT>(o + q, (!valid || wbase + wl < a.fill_words)
                                             ? fill : level(s_w[wave][wl], j));
                }
            }
        }
    }
}
