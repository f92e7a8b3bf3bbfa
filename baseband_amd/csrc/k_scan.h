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
                 bb_frame_rec *recs, uint64_t nframes)
{
    const uint64_t gtid = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
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
    const bool frame_ok = ((bal >> base) & 0xffull) == 0xffull;
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
                   bb_frame_rec *recs, uint64_t nframes)
{
    const uint64_t frame = (uint64_t)blockIdx.x * BB_WAVES_PER_BLOCK + bb_wave();
    if (frame >= nframes) return;                       // wave-uniform
    const int lane = bb_lane();
    const uint64_t off = p.first_offset + frame * (uint64_t)BB_M5B_FRAME;
    const bool whole = off + BB_M5B_FRAME <= nbytes;
    const uint32_t *fw = reinterpret_cast<const uint32_t *>(buf + off);
    uint32_t w = 0;
    if (whole && lane < 4) w = fw[lane];
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
        if (p.frame_rate > 0) {
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
        r.flags = (uint16_t)((sync_ok ? BB_FRAME_OK : 0u) | (all_fill ? BB_FRAME_INVALID : 0u));
        *reinterpret_cast<bb_u4 *>(&recs[frame]) = *reinterpret_cast<const bb_u4 *>(&r);
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
