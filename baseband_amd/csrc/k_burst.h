// Contiguous 2-bit decode with a LOADER wave: the packed bytes of a long work
// item (up to 64 KiB of input = 1 MiB of output) are moved HBM -> LDS by one
// wave with direct-to-LDS loads (global_load_lds_dwordx4: no VGPR round trip),
// while NSTORE other waves expand the item staged before it; two staging
// buffers, one raw s_barrier per item ((a) LDS-direct
// loads, (b) one loader wave feeding store waves through a double-buffered
// stage, (c) 64-128 KiB staged per CU so that HBM sees long read bursts).
//
// Why: the 6 % of the traffic that is reads costs 14 % of the time of
// k_decode_flat_lds, and a streaming reader NEXT to a cache-fed decode costs
// the same (docs/DESIGN_rounds1-3.md 3.3) -- what the memory sees is a steady trickle of
// reads inside a write stream, however the waves issue them.  Here a CU asks
// for its next 64 KiB in one go and then only writes for tens of
// microseconds; with `period` set, all loader waves of the device issue at
// the same ticks of the 100 MHz wall clock, so the reads of the whole chip
// arrive as bursts between which HBM sees stores only.
//
// Work: a launch's frame-slots are cut into PIECES -- a whole payload, or, for
// payloads longer than a staging buffer, a segment of `seg_bytes` -- numbered
// in output order; an ITEM is `kpi` consecutive pieces, whose output is one
// contiguous run.  Piece q of an item is staged at q * fstride + (address &
// 15), so every 16-byte load is aligned in HBM and in LDS; bytes outside a
// payload are never read (ragged ends: dword loads through VGPRs; payloads at
// odd addresses: byte loads).  Invalid pieces (index -1 / out of the buffer)
// are not loaded, their offset entry is -1 and the store waves write fill.
//
// Replaces (reference): the same as k_lds.h -- vdif/payload.py:83-86 with the
// table of :25-63, base/payload.py:314-330, base/frame.py:191-199.
#pragma once
#include "k_flat.h"

#define BB_BURST_MAXK 64            // pieces per item at most
#define BB_BURST_HEAD (4096 + 2 * BB_BURST_MAXK * 4)     // byte table + piece offsets

struct bb_burst_args {
    const uint8_t *buf;
    const int64_t *src;     // [nfs] payload offsets, -1 = fill; may be null
    float         *out;
    const float   *tab;
    uint64_t nfs;           // frame-slots
    uint64_t pbytes;        // payload bytes per frame-slot (multiple of 4)
    uint64_t nseg;          // pieces per frame-slot (1 when kpi > 1)
    uint64_t nitems;
    uint32_t seg_bytes;     // bytes per piece when nseg > 1 (multiple of 256)
    uint32_t kpi;           // pieces per item (1 when nseg > 1)
    uint32_t fstride;       // LDS bytes per piece (multiple of 16, >= piece bytes + 15)
    uint32_t magic;         // floor(2^32 / pbytes) + 1 (kpi > 1): v / pbytes by __umulhi
    uint32_t buf_bytes;     // LDS bytes per staging buffer (multiple of 16)
    uint32_t period;        // loader time slot in wall-clock ticks (10 ns); 0 = issue at once
    int64_t  src0, src_stride;
    float    fill_re, fill_im;
    int32_t  complex_data;
    uint64_t src_lim;
    bb_perm_t perm;         // order of the ITEMS
};

typedef const __attribute__((address_space(1))) void *bb_gptr;
typedef __attribute__((address_space(3))) void *bb_lptr;

template <int BPS, bool NT, int NSTORE, int TAG = 0>
__global__ __launch_bounds__((NSTORE + 1) * BB_WAVE)
void k_decode_flat_burst(bb_burst_args a)
{
    static_assert(BPS == 2, "2-bit samples");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    bb_f4 *s_lut = reinterpret_cast<bb_f4 *>(s_dyn);
    int32_t *s_off = reinterpret_cast<int32_t *>(s_dyn + 4096);      // [2][MAXK]
    uint8_t *s_buf = s_dyn + BB_BURST_HEAD;                            // [2][buf_bytes]
    for (int i = threadIdx.x; i < 256; i += (NSTORE + 1) * BB_WAVE) {
        const uint32_t b = (uint32_t)i;
        s_lut[i] = bb_f4{a.tab[b & 3], a.tab[(b >> 2) & 3], a.tab[(b >> 4) & 3], a.tab[(b >> 6) & 3]};
    }
    __syncthreads();
    const int lane = bb_lane();
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    const bool loader = wave == NSTORE;
    const uint64_t nwork = a.nfs * a.nseg;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    if (blockIdx.x >= a.nitems) return;

    // ---- the loader's work: stage item `it` into buffer `sel`
    auto load_item = [&](uint64_t it, uint32_t sel) {
        const uint64_t w0 = it * a.kpi;
        const uint32_t np = (uint32_t)((nwork - w0 < (uint64_t)a.kpi) ? nwork - w0 : (uint64_t)a.kpi);
        // every piece's source offset with one vector load
        int64_t so_l = -1;
        if ((uint32_t)lane < np) {
            const uint64_t w = w0 + (uint32_t)lane;
            const uint64_t fs = a.nseg == 1 ? w : w / a.nseg;
            so_l = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
        }
        for (uint32_t q = 0; q < np; ++q) {
            const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(so_l & 0xffffffff), q);
            const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)so_l >> 32), q);
            const int64_t so = (int64_t)(((uint64_t)hi32 << 32) | lo32);
            const uint64_t w = w0 + q;
            uint64_t boff = 0, nb = a.pbytes;
            if (a.nseg != 1) {
                const uint64_t fs = w / a.nseg, seg = w - fs * a.nseg;
                boff = seg * a.seg_bytes;
                nb = a.pbytes - boff < (uint64_t)a.seg_bytes ? a.pbytes - boff : (uint64_t)a.seg_bytes;
            }
            if (!bb_src_ok(so, a.src_lim)) {
                if (lane == 0) s_off[sel * BB_BURST_MAXK + q] = -1;
                continue;
            }
            const uint8_t *pp = a.buf + (uint64_t)so + boff;
            const uint32_t a16 = (uint32_t)(reinterpret_cast<uintptr_t>(pp) & 15);
            const uint32_t pbase = sel * a.buf_bytes + q * a.fstride;       // in s_buf
            if (lane == 0) s_off[sel * BB_BURST_MAXK + q] = (int32_t)(pbase + a16);
            if (reinterpret_cast<uintptr_t>(pp) & 3) {
                // a payload at an odd address (repaired files): byte loads
                uint8_t *st8 = s_buf + pbase + a16;
#pragma nounroll
                for (uint32_t i = (uint32_t)lane; i < (uint32_t)nb; i += BB_WAVE) st8[i] = pp[i];
                continue;
            }
            const uint8_t *base = pp - a16;                                 // 16-byte aligned
            const uint32_t lo = a16, hi = a16 + (uint32_t)nb;
            const uint32_t n16 = (hi + 15) >> 4;
#pragma nounroll
            for (uint32_t j0 = 0; j0 < n16; j0 += BB_WAVE) {
                const uint32_t p0 = (j0 + (uint32_t)lane) * 16;
                if (p0 >= lo && p0 + 16 <= hi) {
                    __builtin_amdgcn_global_load_lds((bb_gptr)(base + p0), (bb_lptr)(s_buf + pbase + j0 * 16), 16, 0, 0);
                } else if (p0 < hi && p0 + 16 > lo) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t q4 = p0 + 4 * d;
                        if (q4 >= lo && q4 + 4 <= hi)
                            *reinterpret_cast<uint32_t *>(s_buf + pbase + q4) = *reinterpret_cast<const uint32_t *>(base + q4);
                    }
                }
            }
        }
    };

    // Two separate loops with the same number of barriers: the store waves' loop
    // holds no vector load, so the compiler has nothing to wait for with
    // vmcnt(0) there -- which would also wait for the wave's own stores.
    const uint64_t G = gridDim.x;
    if (loader) {
        uint64_t step = blockIdx.x;
        load_item(bb_perm(a.perm, step), 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (uint32_t n = 0;; ++n) {
            const uint64_t nstep = step + G;
            const bool more = nstep < a.nitems;
            if (more) {
                if (a.period) {
                    // all loader waves of the device issue at the same ticks
                    const uint64_t t = wall_clock64();
                    const uint64_t target = (t / a.period + 1) * a.period;
                    while (wall_clock64() < target) __builtin_amdgcn_s_sleep(8);
                }
                load_item(bb_perm(a.perm, nstep), (n + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (!more) break;
            step = nstep;
        }
        return;
    }
    asm volatile("s_barrier" ::: "memory");
    const bool one = a.kpi == 1;
    for (uint64_t step = blockIdx.x, n = 0; step < a.nitems; step += G, ++n) {
        const uint64_t it = bb_perm(a.perm, step);
        const uint32_t sel = (uint32_t)n & 1;
        const uint64_t w0 = it * a.kpi;
        uint64_t ibytes, ebyte0;            // bytes of the item, its first byte in the output's byte order
        uint32_t pb;                        // bytes per piece
        if (a.nseg == 1) {
            const uint64_t np = nwork - w0 < (uint64_t)a.kpi ? nwork - w0 : (uint64_t)a.kpi;
            pb = (uint32_t)a.pbytes;
            ibytes = np * a.pbytes;
            ebyte0 = w0 * a.pbytes;
        } else {
            const uint64_t fs = w0 / a.nseg, seg = w0 - fs * a.nseg;
            const uint64_t boff = seg * a.seg_bytes;
            ibytes = a.pbytes - boff < (uint64_t)a.seg_bytes ? a.pbytes - boff : (uint64_t)a.seg_bytes;
            pb = (uint32_t)ibytes;
            ebyte0 = fs * a.pbytes + boff;
        }
        float *obase = a.out + ebyte0 * 4;
        const int32_t *off = s_off + sel * BB_BURST_MAXK;
        const uint32_t nchunk = (uint32_t)((ibytes + 255) >> 8);
        for (uint32_t c = (uint32_t)wave; c < nchunk; c += NSTORE) {
            if ((uint64_t)(c + 1) * 256 <= ibytes) {
                // a whole chunk: four independent passes, no branches, so that the
                // twelve LDS reads go out in three batches
                uint32_t vv[4], ii[4];
                int32_t oo[4];
                uint8_t bb[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    vv[p] = c * 256 + 64 * p + (uint32_t)lane;
                    uint32_t q = 0;
                    ii[p] = vv[p];
                    if (!one) { q = __umulhi(vv[p], a.magic); ii[p] = vv[p] - q * pb; }
                    oo[p] = off[q];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) bb[p] = s_buf[oo[p] >= 0 ? (uint32_t)oo[p] + ii[p] : 0u];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bb_f4 t = s_lut[bb[p]];
                    const bool ok = oo[p] >= 0;
                    const bb_f4 val = bb_f4{ok ? t.x : fillv.x, ok ? t.y : fillv.y, ok ? t.z : fillv.z, ok ? t.w : fillv.w};
                    bb_store4<NT>(obase + 4 * (uint64_t)vv[p], val);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t v = c * 256 + 64 * p + (uint32_t)lane;
                    if (v >= (uint32_t)ibytes) continue;
                    uint32_t q = 0, i = v;
                    if (!one) { q = __umulhi(v, a.magic); i = v - q * pb; }
                    const int32_t o = off[q];
                    bb_f4 val = fillv;
                    if (o >= 0) val = s_lut[s_buf[(uint32_t)o + i]];
                    bb_store4<NT>(obase + 4 * (uint64_t)v, val);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
