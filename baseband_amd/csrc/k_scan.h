// Frame-index kernels: header scan (VDIF, Mark 5B) and dense index build.
//
// Replaces (reference, path:line): VDIFHeader.fromfile + verify
// (vdif/header.py:158-186,569-589), the stream-invariant pattern match of
// base/header.py:588-638 as used by locate_frames (base/base.py:181-335),
// VDIFStreamBase._get_index (vdif/base.py:386-390), Mark5BHeader fields
// (mark5b/header.py:60-68), Mark5BStreamBase._get_index
// (mark5b/base.py:206-213), the Mark 5B fill-pattern validity test
// (mark5b/frame.py:62-70), and VDIFFrameSet.fromfile's gather of threads by
// time (vdif/frame.py:176-243).
#pragma once
#include "bb_common.h"

// VDIF: 8 lanes per frame (one header word each), 8 frames per wave; the
// per-word invariant tests are combined with one wave-wide ballot.
__global__ __launch_bounds__(BB_BLOCK)
void k_vdif_scan(const uint8_t *buf, uint64_t nbytes, bb_vdif_scan_params p,
                 bb_frame_rec *recs, uint64_t nframes, int64_t *fill, uint64_t fill_n)
{
    const uint64_t gtid = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    // (bb_vdif_read_window: the dense index this window's records are scattered into
    // by the NEXT launch is pre-set to -1 here instead of by a memset of its own)
    if (fill)
        for (uint64_t i = gtid; i < fill_n; i += (uint64_t)gridDim.x * BB_BLOCK) fill[i] = -1;
    const uint64_t frame = gtid >> 3;
    const int wi = (int)(gtid & 7);
    const int lane = bb_lane();
    const int nwords = (int)(p.header_nbytes >> 2);
    const uint64_t off = p.first_offset + frame * (uint64_t)p.frame_nbytes;

    uint32_t pat = 0, msk = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (wi == k) { pat = p.pattern[k]; msk = p.mask[k]; }

    uint32_t w = 0;
    bool ok = true;
    if (frame < nframes && wi < nwords) {
        if (off + 4u * wi + 4u <= nbytes) {
            w = *reinterpret_cast<const uint32_t *>(buf + off + 4u * wi);
            ok = ((w ^ pat) & msk) == 0;
        } else {
            ok = false;                     // truncated header
        }
    }
    const unsigned long long bal = __ballot(ok);
    const int base = lane & ~7;
    // a frame counts only when all of it lies inside the buffer: the decode
    // kernels take payload offsets from these records without a length check
    const bool frame_ok = ((bal >> base) & 0xffull) == 0xffull
                          && off + (uint64_t)p.frame_nbytes <= nbytes;
    const uint32_t w1 = (uint32_t)__shfl((int)w, base + 1);
    const uint32_t w3 = (uint32_t)__shfl((int)w, base + 3);
    if (wi == 0 && frame < nframes) {
        const int32_t seconds = (int32_t)(w & 0x3fffffffu);
        const uint32_t invalid = w >> 31;
        const int32_t frame_nr = (int32_t)(w1 & 0x00ffffffu);
        const int32_t thread_id = (int32_t)((w3 >> 16) & 0x3ffu);
        int64_t tidx;
        if (p.frame_rate > 0)
            tidx = (int64_t)(seconds - p.ref_seconds) * p.frame_rate
                   + (frame_nr - p.ref_frame_nr);
        else
            tidx = (int64_t)frame;
        // Frame sets as the reference forms them (VDIFFrameSet.fromfile, vdif/frame.py:201-243):
        // the frames that follow a set's first header with the SAME frame_nr are that
        // set, whatever their seconds say; the set is placed by its first header.
        if (p.set_nframes > 1 && p.frame_rate > 0) {
            const uint64_t lead = frame - frame % (uint64_t)p.set_nframes;
            const uint64_t loff = p.first_offset + lead * (uint64_t)p.frame_nbytes;
            if (lead != frame && loff + 8 <= nbytes) {
                const uint32_t l0 = *reinterpret_cast<const uint32_t *>(buf + loff);
                const uint32_t l1 = *reinterpret_cast<const uint32_t *>(buf + loff + 4);
                if ((int32_t)(l1 & 0x00ffffffu) == frame_nr)
                    tidx = (int64_t)((int32_t)(l0 & 0x3fffffffu) - p.ref_seconds) * p.frame_rate
                           + (frame_nr - p.ref_frame_nr);
            }
        }
        if (tidx > 0x7fffffffll) tidx = 0x7fffffffll;
        if (tidx < -0x7fffffffll) tidx = -0x7fffffffll;
        bb_frame_rec r;
        r.payload_offset = (int64_t)(off + p.header_nbytes);
        r.time_index = (int32_t)tidx;
        r.thread_id = (int16_t)thread_id;
        r.flags = (uint16_t)((frame_ok ? BB_FRAME_OK : 0u) | (invalid ? BB_FRAME_INVALID : 0u));
        *reinterpret_cast<bb_u4 *>(&recs[frame]) = *reinterpret_cast<const bb_u4 *>(&r);
    }
}

// Unaligned little-endian dword at byte position pos, assembled from the two
// aligned dwords around it (positions found by the byte-granular locate
// kernel need not be aligned once bytes went missing from a file).
__device__ __forceinline__ uint32_t bb_load_u32_any(const uint8_t *buf, uint64_t nbytes, uint64_t pos)
{
    const uint64_t a = pos & ~3ull;
    const uint32_t sh = (uint32_t)(pos & 3) * 8;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(buf + a);
    const uint32_t lo = (a + 4 <= nbytes) ? w[0] : 0u;
    if (sh == 0) return lo;
    const uint32_t hi = (a + 8 <= nbytes) ? w[1] : 0u;
    return (lo >> sh) | (hi << (32 - sh));
}

__device__ __forceinline__ bool bb_vdif_header_at(const uint8_t *buf, uint64_t nbytes,
                                                  const bb_vdif_scan_params &p, uint64_t pos)
{
    const int nwords = (int)(p.header_nbytes >> 2);
    if (pos + p.header_nbytes > nbytes) return false;
    // all nine aligned dwords that hold the (possibly unaligned) header are
    // requested before any of them is looked at: ONE memory latency per
    // header.  (Testing word by word with early exits made every true frame
    // of a search cost eight dependent loads, twice -- k_vdif_locate 2.7 ms
    // for 8 GiB against 1.6 ms for the Mark 5B search.)
    const uint64_t a = pos & ~3ull;
    const uint32_t sh = (uint32_t)(pos & 3) * 8;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(buf + a);
    uint32_t d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
        d[k] = (k <= nwords && a + 4 * (uint64_t)k + 4 <= nbytes) ? w[k] : 0u;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t v = sh ? (d[k] >> sh) | (d[k + 1] << (32 - sh)) : d[k];
        if (k < nwords) ok = ok && (((v ^ p.pattern[k]) & p.mask[k]) == 0);
    }
    return ok;
}

// ---- byte-granular frame search -------------------------------------------
// Skeleton shared by the VDIF / Mark 5B / Mark 4 searches (the
// masked-pattern search of locate_frames, base/base.py:181-335).  A lane loads
// 16 aligned bytes (four dwords; the dword behind them comes from the next
// lane) and forms the little-endian dword at each of its 16 byte positions with
// one v_alignbyte each: the file is read ONCE, 16 bytes per lane, and a position
// costs three or four ALU operations.  `probe(v)` is the format's test on the
// most selective dword of its header pattern -- random bytes pass it once in
// 2^29 .. 2^32 positions -- and only those candidates go through `confirm(pos)`,
// the complete test (all header words, frame fits, a header one frame later).
// Round 1 tested every byte position in full, with every header word rebuilt
// from two aligned loads: 0.61 TB/s of file bytes.
// Confirmed positions are collected per workgroup in LDS and appended to the
// global list with ONE atomic per workgroup: a million frames reporting through
// a single global counter took 12 of the kernel's 13 ms (one word takes about 88
// returning atomics per microsecond, MI355X_MICROARCH.md price list "dequeue").
#define BB_LOCATE_LOCAL 1024
#ifndef BB_LOCATE_U
#define BB_LOCATE_U 4
#endif
template <class Probe, class Confirm>
__device__ __forceinline__ void bb_locate_sweep(const uint8_t *buf, uint64_t nbytes, uint64_t q_end,
                                                 int64_t *out, uint64_t cap, unsigned long long *count,
                                                 Probe probe, Confirm confirm)
{
    __shared__ int64_t s_found[BB_LOCATE_LOCAL];
    __shared__ uint64_t s_cand[BB_LOCATE_LOCAL];
    __shared__ uint32_t s_n, s_nc;
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) { s_n = 0; s_nc = 0; }
    __syncthreads();
    // byte positions q in [0, q_end) are probed; q_end + 4 <= nbytes
    const uint64_t nchunk = (q_end + 15) / 16;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(buf);
    const uint64_t ndw = nbytes / 4;
    const int lane = bb_lane();
    // BB_LOCATE_U chunks of 16 bytes per lane and iteration, all their loads issued
    // before the first probe: with one load in flight per lane the sweep was
    // latency bound -- 32 waves x 1 KiB per CU in flight = 8 MiB on the chip, at 2
    // us per load 4.2 TB/s, which is what it measured (4.0-4.3; round 4)
    constexpr int U = BB_LOCATE_U;
    // A workgroup takes U x 4 KiB in ONE piece per iteration (a lane's U chunks lie 4 KiB apart)
    // and fetches them with nontemporal loads: a read-only sweep in that shape reaches 0.87-0.90
    // of the peak where plain loads that lie a whole grid apart reach 0.80
    // (tools/experiments/read_probe.cpp, profiles/r06ci_read_probe.log; round 6).
    const uint64_t stride = BB_BLOCK;                               // between a lane's chunks
    const uint64_t step = (uint64_t)gridDim.x * BB_BLOCK * U;       // between a workgroup's iterations
    for (uint64_t j0 = (uint64_t)blockIdx.x * BB_BLOCK * U + threadIdx.x; j0 < nchunk + (BB_WAVE - 1);
         j0 += step) {
        // (whole waves stay in the loop: the shuffle below needs every lane)
        bb_u4 dv[U];
        uint32_t tail[U];
        // Everything this wave touches in this iteration lies inside the buffer (all
        // iterations but the last ones): loads with nothing between them.  (With the
        // bounds tests of the ragged end in the way the compiler waited for the first
        // chunk before it issued the second: round 5.)
        const uint64_t wave_last = (j0 - (uint64_t)lane) + (BB_WAVE - 1) + (uint64_t)(U - 1) * stride;
        if (4 * wave_last + 5 <= ndw) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                dv[u] = __builtin_nontemporal_load(reinterpret_cast<const bb_u4 *>(w + 4 * (j0 + (uint64_t)u * stride)));
            // (the dword after a lane's sixteen bytes: the next lane's first -- every lane
            // fetches its own, out of the lines the loads above bring in: no shuffle, no
            // lane that differs)
#pragma unroll
            for (int u = 0; u < U; ++u)
                tail[u] = w[4 * (j0 + (uint64_t)u * stride) + 4];
        } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t j = j0 + (uint64_t)u * stride;
            const uint64_t dw0 = 4 * j;
            dv[u] = bb_u4{0u, 0u, 0u, 0u};
            tail[u] = 0u;
            if (j < nchunk + (BB_WAVE - 1)) {
                if (dw0 + 4 <= ndw) dv[u] = *reinterpret_cast<const bb_u4 *>(w + dw0);
                else {
                    if (dw0 < ndw) dv[u].x = w[dw0];
                    if (dw0 + 1 < ndw) dv[u].y = w[dw0 + 1];
                    if (dw0 + 2 < ndw) dv[u].z = w[dw0 + 2];
                }
                if (lane == BB_WAVE - 1 && dw0 + 4 < ndw) tail[u] = w[dw0 + 4];
            }
            const uint32_t nx = (uint32_t)__shfl_down((int)dv[u].x, 1);
            if (lane != BB_WAVE - 1) tail[u] = nx;
        }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t j = j0 + (uint64_t)u * stride;
            const bb_u4 d = dv[u];
            const uint32_t nx = tail[u];
            if (j >= nchunk) continue;
            const uint32_t dd[5] = {d.x, d.y, d.z, d.w, nx};
            // `probe` gives the MISMATCHING bits of a dword (0 = it is the pattern): the
            // minimum over the sixteen positions is zero iff one of them hits -- three
            // operations and a third of a v_min3 per position instead of a compare and
            // two mask updates each; which positions hit is worked out only then
            uint32_t worst = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t m0 = probe(dd[k]);
                const uint32_t m1 = probe(__builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 1));
                const uint32_t m2 = probe(__builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 2));
                const uint32_t m3 = probe(__builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 3));
                const uint32_t a01 = m0 < m1 ? m0 : m1, a23 = m2 < m3 ? m2 : m3;
                const uint32_t a = a01 < a23 ? a01 : a23;
                worst = worst < a ? worst : a;
            }
            uint32_t hits = 0;                              // bit 4k+s: position 16j + 4k + s
            if (worst == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (probe(dd[k]) == 0) hits |= 1u << (4 * k);
                    if (probe(__builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 1)) == 0) hits |= 2u << (4 * k);
                    if (probe(__builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 2)) == 0) hits |= 4u << (4 * k);
                    if (probe(__builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 3)) == 0) hits |= 8u << (4 * k);
                }
            }
            while (hits) {                                  // rare
                const int b = __ffs((int)hits) - 1;
                hits &= hits - 1;
                const uint64_t q = 16 * j + (uint64_t)b;
                if (q >= q_end) continue;
                // candidates are parked: confirming one here would stall the whole
                // wave on two scattered header fetches (a real frame's header and
                // the one a frame later) while 63 lanes wait; after the sweep the
                // workgroup confirms all of its candidates at once, one per thread,
                // their fetches in flight together
                const uint32_t c = atomicAdd(&s_nc, 1u);
                if (c < BB_LOCATE_LOCAL) { s_cand[c] = q; continue; }
                const int64_t pos = confirm(q);             // (list full: on the spot, to the global list)
                if (pos < 0) continue;
                const unsigned long long g = atomicAdd(count, 1ull);
                if (g < cap) out[g] = pos;
            }
        }
    }
    __syncthreads();
    const uint32_t nc = s_nc < BB_LOCATE_LOCAL ? s_nc : BB_LOCATE_LOCAL;
    for (uint32_t i = threadIdx.x; i < nc; i += BB_BLOCK) {
        const int64_t pos = confirm(s_cand[i]);
        if (pos >= 0) s_found[atomicAdd(&s_n, 1u)] = pos;
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (threadIdx.x == 0 && n) s_base = atomicAdd(count, (unsigned long long)n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += BB_BLOCK)
        if (s_base + i < cap) out[s_base + i] = s_found[i];
}

// VDIF: position pos is reported when the stream-invariant pattern matches
// there, the whole frame fits in the buffer, and another header sits exactly
// one frame later (or two, when the next header is damaged in place; for the
// last frame in the buffer: one earlier) -- the `check` logic of locate_frames
// (base/base.py:181-335) as used by VDIF's _bad_frame recovery
// (vdif/base.py:536-755).  The probe is header word 2 (version, lg2_nchan,
// frame_length: 8 bytes into the header).  Matches are appended unordered.
__global__ __launch_bounds__(BB_BLOCK)
void k_vdif_locate(const uint8_t *buf, uint64_t nbytes, bb_vdif_scan_params p,
                   int64_t *out, uint64_t cap, unsigned long long *count)
{
    // word 2 of a frame that fits lies at q = pos + 8 <= nbytes - frame_nbytes + 8
    const uint64_t q_end = nbytes - p.frame_nbytes + 8 + 1;
    const uint32_t pat = p.pattern[2], msk = p.mask[2];
    auto confirm = [&](uint64_t q) -> int64_t {
            if (q < 8) return -1;
            const uint64_t pos = q - 8;
            if (pos + p.frame_nbytes > nbytes) return -1;
            const uint64_t next = pos + p.frame_nbytes;
            // (this header and the one a frame later are fetched together)
            const bool here = bb_vdif_header_at(buf, nbytes, p, pos);
            const bool there = next + p.header_nbytes <= nbytes && bb_vdif_header_at(buf, nbytes, p, next);
            if (!here) return -1;
            bool ok;
            if (next + p.header_nbytes <= nbytes) {
                ok = there;
                // the following header may be damaged in place (no bytes lost):
                // then the one after it is still where the stride says
                if (!ok && next + p.frame_nbytes + p.header_nbytes <= nbytes)
                    ok = bb_vdif_header_at(buf, nbytes, p, next + p.frame_nbytes);
            } else {
                ok = pos < p.frame_nbytes || bb_vdif_header_at(buf, nbytes, p, pos - p.frame_nbytes);
            }
            return ok ? (int64_t)pos : -1;
        };
    // (one instantiation of the sweep only: a second one for streams whose
    // word 2 is invariant in all 32 bits -- an equality probe -- doubled the
    // kernel's LDS and ran 25 % slower, profiles/r02am_locate.txt)
    bb_locate_sweep(buf, nbytes, q_end, out, cap, count,
                    [&](uint32_t v) { return (v ^ pat) & msk; }, confirm);
}

// Header scan at explicit (possibly unaligned) frame offsets: same record as
// k_vdif_scan, one lane per frame.
__global__ __launch_bounds__(BB_BLOCK)
void k_vdif_scan_at(const uint8_t *buf, uint64_t nbytes, bb_vdif_scan_params p,
                    const int64_t *offsets, bb_frame_rec *recs, uint64_t nframes)
{
    const uint64_t i = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= nframes) return;
    const uint64_t off = (uint64_t)offsets[i];
    const bool ok = bb_vdif_header_at(buf, nbytes, p, off) && off + p.frame_nbytes <= nbytes;
    const uint32_t w0 = bb_load_u32_any(buf, nbytes, off);
    const uint32_t w1 = bb_load_u32_any(buf, nbytes, off + 4);
    const uint32_t w3 = bb_load_u32_any(buf, nbytes, off + 12);
    const int32_t seconds = (int32_t)(w0 & 0x3fffffffu);
    const int32_t frame_nr = (int32_t)(w1 & 0x00ffffffu);
    int64_t tidx = p.frame_rate > 0
        ? (int64_t)(seconds - p.ref_seconds) * p.frame_rate + (frame_nr - p.ref_frame_nr)
        : (int64_t)i;
    if (tidx > 0x7fffffffll) tidx = 0x7fffffffll;
    if (tidx < -0x7fffffffll) tidx = -0x7fffffffll;
    bb_frame_rec r;
    r.payload_offset = (int64_t)(off + p.header_nbytes);
    r.time_index = (int32_t)tidx;
    r.thread_id = (int16_t)((w3 >> 16) & 0x3ffu);
    r.flags = (uint16_t)((ok ? BB_FRAME_OK : 0u) | ((w0 >> 31) ? BB_FRAME_INVALID : 0u));
    *reinterpret_cast<bb_u4 *>(&recs[i]) = *reinterpret_cast<const bb_u4 *>(&r);
}

__device__ __forceinline__ int bb_bcd_decode(uint32_t v, int ndigit)
{
    int r = 0, m = 1;
    for (int i = 0; i < ndigit; ++i) { r += (int)((v >> (4 * i)) & 0xf) * m; m *= 10; }
    return r;
}

// Mark 5B: one wave per 10016-byte frame.  Lanes 0-3 hold the header words;
// the whole wave then tests payload words against the 0x11223344 fill
// pattern, leaving the loop at the first wave-wide mismatch (for real data
// that is the first iteration: 256 bytes).
#define BB_M5B_FRAME 10016u
#define BB_M5B_PAYLOAD_WORDS 2500u
__global__ __launch_bounds__(BB_BLOCK)
void k_mark5b_scan(const uint8_t *buf, uint64_t nbytes, bb_mark5b_scan_params p,
                   const int64_t *offsets, bb_frame_rec *recs, uint64_t nframes)
{
    const uint64_t frame = (uint64_t)blockIdx.x * BB_WAVES_PER_BLOCK + bb_wave();
    if (frame >= nframes) return;                       // wave-uniform
    const int lane = bb_lane();
    // fixed stride, or explicit (possibly odd) offsets from bb_mark5b_locate
    const uint64_t off = offsets ? (uint64_t)offsets[frame]
                                 : p.first_offset + frame * (uint64_t)BB_M5B_FRAME;
    const bool whole = off + BB_M5B_FRAME <= nbytes;
    const uint32_t *fw = reinterpret_cast<const uint32_t *>(buf + off);
    uint32_t w = 0;
    if (whole && lane < 4) w = bb_load_u32_any(buf, nbytes, off + 4 * lane);
    const uint32_t w0 = (uint32_t)__shfl((int)w, 0);
    const uint32_t w1 = (uint32_t)__shfl((int)w, 1);
    const uint32_t w2 = (uint32_t)__shfl((int)w, 2);
    const bool sync_ok = whole && w0 == 0xABADDEEDu;

    bool all_fill = whole;
    if (whole) {
        for (uint32_t i = lane; i < BB_M5B_PAYLOAD_WORDS + 63u; i += 64) {
            const bool mine = (i < BB_M5B_PAYLOAD_WORDS) ? (fw[4 + i] == 0x11223344u) : true;
            if (!__all(mine)) { all_fill = false; break; }
        }
    }
    if (lane == 0) {
        const int32_t frame_nr = (int32_t)(w1 & 0x7fffu);
        const int jday = bb_bcd_decode(w2 >> 20, 3);
        const int secs = bb_bcd_decode(w2 & 0xfffffu, 5);
        const int32_t seconds = jday * 86400 + secs;
        int64_t tidx;
        if (p.frame_rate > 0 && !p.by_position) {
            int64_t ds = (int64_t)seconds - p.ref_seconds;
            // jday is modulo 1000 days (mark5b/header.py:235-262): unwrap
            if (ds < -500ll * 86400) ds += 1000ll * 86400;
            tidx = ds * p.frame_rate + (frame_nr - p.ref_frame_nr);
        } else {
            tidx = (int64_t)frame;
        }
        if (tidx > 0x7fffffffll) tidx = 0x7fffffffll;
        if (tidx < -0x7fffffffll) tidx = -0x7fffffffll;
        bb_frame_rec r;
        r.payload_offset = (int64_t)(off + 16);
        r.time_index = (int32_t)tidx;
        r.thread_id = 0;
        r.flags = (uint16_t)(((p.by_position ? whole : sync_ok) ? BB_FRAME_OK : 0u) | (all_fill ? BB_FRAME_INVALID : 0u));
        *reinterpret_cast<bb_u4 *>(&recs[frame]) = *reinterpret_cast<const bb_u4 *>(&r);
    }
}

// CRC-16 (x^16 + x^15 + x^2 + 1) of the 48 time-code bits of header words 2
// and 3, compared with the 16 bits stored below them (mark5b/header.py:28-31,
// used by find_header: mark5b/base.py:136-155).
__device__ __forceinline__ bool bb_mark5b_crc_ok(uint32_t w2, uint32_t w3)
{
    const uint64_t stream = ((uint64_t)w2 << 32) | w3;  // 48 data bits, then 16 CRC bits
    uint32_t reg = 0;
#pragma unroll 8
    for (int i = 63; i >= 16; --i) {
        const uint32_t bit = (uint32_t)(stream >> i) & 1u;
        const uint32_t top = (reg >> 15) & 1u;
        reg = (reg << 1) & 0xffffu;
        if (top ^ bit) reg ^= 0x8005u;
    }
    return reg == (uint32_t)(w3 & 0xffffu);
}

// Byte-granular Mark 5B frame search: position p holds a frame
// when the sync word sits at p, the whole frame fits in the buffer, the time
// code passes its CRC, and -- if four more bytes fit there -- another sync word
// sits exactly one frame later: locate_frames with check=1 plus the CRC gate
// of Mark5BFileReader.find_header (base/base.py:181-335, mark5b/base.py:136-155),
// which is what the reference's _bad_frame recovery accepts
// (base/base.py:1127-1219).  Matches are appended unordered.
__global__ __launch_bounds__(BB_BLOCK)
void k_mark5b_locate(const uint8_t *buf, uint64_t nbytes, uint32_t w1_pattern, uint32_t w1_mask,
                     int64_t *out, uint64_t cap, unsigned long long *count)
{
    const uint64_t q_end = nbytes - BB_M5B_FRAME + 1;       // the sync word is the probe
    bb_locate_sweep(buf, nbytes, q_end, out, cap, count,
        [&](uint32_t v) { return v ^ 0xABADDEEDu; },
        [&](uint64_t pos) -> int64_t {
            // (word 1 under the mask: the bits that are the same in every header of ONE stream --
            // the user word, mark5b/header.py:70-73 -- here and one frame later, where the
            // reference's locate_frames compares the whole pattern of header0 again)
            if (w1_mask && ((bb_load_u32_any(buf, nbytes, pos + 4) ^ w1_pattern) & w1_mask)) return -1;
            const uint64_t next = pos + BB_M5B_FRAME;
            if (next + 4 <= nbytes && bb_load_u32_any(buf, nbytes, next) != 0xABADDEEDu) return -1;
            if (w1_mask && next + 8 <= nbytes
                && ((bb_load_u32_any(buf, nbytes, next + 4) ^ w1_pattern) & w1_mask)) return -1;
            if (!bb_mark5b_crc_ok(bb_load_u32_any(buf, nbytes, pos + 8),
                                  bb_load_u32_any(buf, nbytes, pos + 12))) return -1;
            return (int64_t)pos;
        });
}

// A pass that only READS [buf, buf + nbytes): what it brings into the 256 MiB memory-side cache the
// decode launch that follows finds there (bb_touch; round 6).  PLAIN loads -- nontemporal ones go
// past that cache -- one 16-byte load per lane, which is the shape plain loads are fastest in
// (tools/experiments/read_probe.cpp: 0.88-0.90 of the peak; four in flight: 0.81).
__global__ __launch_bounds__(BB_BLOCK)
void k_touch(const bb_u4 *in, uint64_t nchunk, uint32_t *sink)
{
    const uint64_t j = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    if (j >= nchunk) return;
    const bb_u4 v = in[j];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x5bd1e995u && sink) sink[0] = v.x;      // (never with sink = NULL: keeps the load)
}

// Verification of a window's scan records in one launch: counts the records
// that are not BB_FRAME_OK, or -- for the first `nstrict` of them -- whose time
// index is not first_index + i / recs_per_index (frames out of place).  The
// records after `nstrict` are the look-ahead header behind the request: it only
// has to be a header (base/base.py:1083-1125).  The count is ADDED to *nbad.
__global__ __launch_bounds__(BB_BLOCK)
void k_verify_records(const bb_frame_rec *recs, uint64_t nrecs, int32_t first_index,
                      uint32_t recs_per_index, uint64_t nstrict, uint32_t *nbad)
{
    const uint64_t i = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    bool bad = false;
    if (i < nrecs) {
        const bb_u4 raw = *reinterpret_cast<const bb_u4 *>(&recs[i]);
        bb_frame_rec r;
        *reinterpret_cast<bb_u4 *>(&r) = raw;
        bad = !(r.flags & BB_FRAME_OK)
              || (i < nstrict && r.time_index != first_index + (int32_t)(i / recs_per_index));
    }
    const unsigned long long m = __ballot(bad);
    if (bb_lane() == 0 && m) atomicAdd(nbad, (uint32_t)__popcll(m));
}

// k_build_index and k_verify_records in ONE launch (the read_window entry points:
// a mid-size read is five launches otherwise, and the three small ones between two
// decodes cost 20-30 us of a 650 us read; round 4).  `nbad` may be null.
__global__ __launch_bounds__(BB_BLOCK)
void k_index_verify(const bb_frame_rec *recs, uint64_t nrecs, const int16_t *thread_slot, int nslot,
                    int64_t *src, uint64_t nframes_out, uint32_t recs_per_index, uint64_t nstrict, uint32_t *nbad)
{
    const uint64_t i = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    bool bad = false;
    if (i < nrecs) {
        const bb_u4 raw = *reinterpret_cast<const bb_u4 *>(&recs[i]);
        bb_frame_rec r;
        *reinterpret_cast<bb_u4 *>(&r) = raw;
        bad = !(r.flags & BB_FRAME_OK) || (i < nstrict && r.time_index != (int32_t)(i / recs_per_index));
        bool put = (r.flags & BB_FRAME_OK) && r.time_index >= 0 && (uint64_t)r.time_index < nframes_out
                   && !(r.flags & BB_FRAME_INVALID);
        int slot = 0;
        if (put && thread_slot) {
            slot = thread_slot[r.thread_id & 0x3ff];
            put = slot >= 0 && slot < nslot;
        }
        if (put) src[(uint64_t)r.time_index * nslot + slot] = r.payload_offset;
    }
    if (nbad) {
        const unsigned long long m = __ballot(bad);
        if (bb_lane() == 0 && m) atomicAdd(nbad, (uint32_t)__popcll(m));
    }
}

// Scatter scan records into the dense output-ordered source table (which the
// caller's launch wrapper pre-fills with -1).
__global__ __launch_bounds__(BB_BLOCK)
void k_build_index(const bb_frame_rec *recs, uint64_t nrecs,
                   const int16_t *thread_slot, int nslot,
                   int64_t *src, uint64_t nframes_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    if (i >= nrecs) return;
    const bb_u4 raw = *reinterpret_cast<const bb_u4 *>(&recs[i]);
    bb_frame_rec r;
    *reinterpret_cast<bb_u4 *>(&r) = raw;
    if (!(r.flags & BB_FRAME_OK)) return;
    if (r.time_index < 0 || (uint64_t)r.time_index >= nframes_out) return;
    int slot = 0;
    if (thread_slot) {
        slot = thread_slot[r.thread_id & 0x3ff];
        if (slot < 0 || slot >= nslot) return;
    }
    if (r.flags & BB_FRAME_INVALID) return;             // stays -1 -> fill
    src[(uint64_t)r.time_index * nslot + slot] = r.payload_offset;
}
