// int8 (re, im) pairs -> complex64 with an axis permutation, through LDS.
//
// Replaces (reference, path:line): GUPPIPayload._decode, channels-first
//   words.reshape(nchan, -1)... .T.reshape(-1, npol, nchan)   (guppi/payload.py:90-96)
// and time-first .reshape(-1, nchan, npol).transpose(0, 2, 1)  (guppi/payload.py:97-102),
// MKBFPayload's np.moveaxis(words["heaps"], -1, 1)             (dada/payload.py:76-79),
// plus the int8 -> float32 casts (guppi/payload.py:13-14, dada/payload.py:13-14).
//
// Output is always (time, pol, chan) complex64.  The input address of element
// (t, p, c), in 2-byte elements, is
//     (t / tb) * sh + (t % tb) * st + p * sp + c * sc
//   layout 0  GUPPI channels-first (chan, time, pol):  st = npol, sp = 1, sc = T*npol
//   layout 1  MKBF heaps (heap, pol, chan, 256):       tb = 256, sh = npol*nchan*256,
//                                                       st = 1, sp = nchan*256, sc = 256
//   layout 2  GUPPI time-first (time, chan, pol):      st = nchan*npol, sp = 1, sc = npol
//
// A workgroup moves a tile of tt times x tc channels x all pols.  Phase 1
// walks the tile in INPUT order (lanes along the contiguous input axis:
// 128-byte wave loads) and drops each 2-byte element into LDS at its OUTPUT
// position [t][p][c]; the row pitch (tc rounded up to an odd number of dwords) keeps
// the strided writes off the same bank.  Phase 2 reads LDS linearly: a lane
// takes one dword (two elements) and stores one float4, so output rows are
// written in contiguous 16-byte pieces (512 B per 64-channel row).
#pragma once
#include "bb_common.h"

struct bb_tiled_args {
    const uint8_t *buf;
    const int64_t *src;
    float         *out;
    uint64_t nframes;
    uint64_t t_lo, t_hi;        // local time range decoded from every frame
    uint64_t tb, sh, st, sp, sc;
    int64_t  src0, src_stride;
    uint32_t npol, nchan;
    uint32_t tt, tc;            // tile: times, channels
    uint32_t tcp;               // LDS row pitch in elements: even, odd number of dwords
    uint32_t ntt, nct;          // tiles per frame along time / channel
    float    fill_re, fill_im;
    uint64_t src_lim;       // offsets outside [0, src_lim) decode as fill (bb_src_ok)
    bb_perm_t perm;         // work order (bb_common.h)
    // a reader subset folded into the decode (k_decode_i8_xpose only): `npol`
    // polarisations starting at `pf` of the `nps` stored ones, and channel c of
    // the output = stored channel cmap[c] (null: c)
    uint32_t nps, pf;
    const int32_t *cmap;
};

template <int LAYOUT, bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_i8_tiled(bb_tiled_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t s_tile[];
    const uint32_t tcp = a.tcp;                         // padded row pitch (elements)
    const uint32_t npol = a.npol, tc = a.tc, tt = a.tt;
    const uint64_t rows_out = a.t_hi - a.t_lo;          // output rows per frame
    const uint64_t rowlen = (uint64_t)npol * a.nchan * 2;   // floats per output time
    const uint64_t nwork = a.nframes * a.ntt * a.nct;
    const uint32_t tile_elems = tt * npol * tc;

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / ((uint64_t)a.ntt * a.nct);
        const uint32_t rem = (uint32_t)(work - f * a.ntt * a.nct);
        const uint32_t ti = rem / a.nct, ci = rem - ti * a.nct;
        const uint64_t t0 = a.t_lo + (uint64_t)ti * tt;
        const uint32_t nt_tile = (uint32_t)((a.t_hi - t0 < tt) ? a.t_hi - t0 : tt);
        const uint32_t c0 = ci * tc;
        const uint32_t nc_tile = (a.nchan - c0 < tc) ? a.nchan - c0 : tc;
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint16_t *in = reinterpret_cast<const uint16_t *>(a.buf + (valid ? so : 0));

        if (valid) {
            // phase 1: input order -> LDS at output position [t][p][c]
            const int lane = bb_lane(), wave = bb_wave();
            if (LAYOUT == 0) {
                // (c, t, p): within a channel the (t, p) run is contiguous and
                // its index IS the LDS row.  One wave per channel, lanes along
                // the run, two elements (one dword) per lane when aligned.
                const uint32_t run = nt_tile * npol;
                for (uint32_t c = wave; c < nc_tile; c += BB_WAVES_PER_BLOCK) {
                    const uint64_t off = (uint64_t)(c0 + c) * a.sc + t0 * npol;
                    const uint16_t *row = in + off;
                    if (((uintptr_t)row & 3) == 0) {
                        for (uint32_t i = lane * 2; i < run; i += 128) {
                            if (i + 1 < run) {
                                const uint32_t w = *reinterpret_cast<const uint32_t *>(row + i);
                                s_tile[i * tcp + c] = (uint16_t)(w & 0xffff);
                                s_tile[(i + 1) * tcp + c] = (uint16_t)(w >> 16);
                            } else {
                                s_tile[i * tcp + c] = row[i];
                            }
                        }
                    } else {
                        for (uint32_t i = lane; i < run; i += 64) s_tile[i * tcp + c] = row[i];
                    }
                }
            } else if (LAYOUT == 1 && (t0 % a.tb) + nt_tile <= a.tb) {
                // (heap, p, c, t): a (p, c) pair has contiguous times inside a heap
                const uint64_t hbase = (t0 / a.tb) * a.sh + (t0 % a.tb) * a.st;
                for (uint32_t pc = wave; pc < npol * nc_tile; pc += BB_WAVES_PER_BLOCK) {
                    const uint32_t p = pc / nc_tile, c = pc - p * nc_tile;
                    const uint16_t *row = in + hbase + (uint64_t)p * a.sp + (uint64_t)(c0 + c) * a.sc;
                    if (((uintptr_t)row & 3) == 0) {
                        for (uint32_t tl = lane * 2; tl < nt_tile; tl += 128) {
                            if (tl + 1 < nt_tile) {
                                const uint32_t w = *reinterpret_cast<const uint32_t *>(row + tl);
                                s_tile[(tl * npol + p) * tcp + c] = (uint16_t)(w & 0xffff);
                                s_tile[((tl + 1) * npol + p) * tcp + c] = (uint16_t)(w >> 16);
                            } else {
                                s_tile[(tl * npol + p) * tcp + c] = row[tl];
                            }
                        }
                    } else {
                        for (uint32_t tl = lane; tl < nt_tile; tl += 64)
                            s_tile[(tl * npol + p) * tcp + c] = row[tl];
                    }
                }
            } else {
                for (uint32_t i = threadIdx.x; i < tile_elems; i += BB_BLOCK) {
                    uint32_t tl, p, c;
                    if (LAYOUT == 1) {              // (p, c, t), tile crosses a heap
                        p = i / (tc * tt);
                        const uint32_t r = i - p * tc * tt;
                        c = r / tt; tl = r - c * tt;
                    } else {                        // (t, c, p)
                        tl = i / (tc * npol);
                        const uint32_t r = i - tl * tc * npol;
                        c = r / npol; p = r - c * npol;
                    }
                    if (tl < nt_tile && c < nc_tile) {
                        const uint64_t t = t0 + tl;
                        const uint64_t off = (t / a.tb) * a.sh + (t % a.tb) * a.st
                                             + (uint64_t)p * a.sp + (uint64_t)(c0 + c) * a.sc;
                        s_tile[(tl * npol + p) * tcp + c] = in[off];
                    }
                }
            }
        }
        __syncthreads();
        // phase 2: LDS rows -> global, one dword (2 elements) -> one float4
        const uint32_t pairs = (tc + 1) / 2;
        const uint32_t nrows = nt_tile * npol;
        float *obase = a.out + (f * rows_out + (t0 - a.t_lo)) * rowlen + (uint64_t)c0 * 2;
        for (uint32_t j = threadIdx.x; j < nrows * pairs; j += BB_BLOCK) {
            const uint32_t row = j / pairs;
            const uint32_t c = (j - row * pairs) * 2;
            if (c >= nc_tile) continue;
            float *o = obase + (uint64_t)row * a.nchan * 2 + (uint64_t)c * 2;
            float v0, v1, v2 = 0.f, v3 = 0.f;
            const bool two = c + 1 < nc_tile;
            if (valid) {
                // tcp and c are even: the two elements form one aligned dword
                const uint32_t e = *reinterpret_cast<const uint32_t *>(&s_tile[row * tcp + c]);
                v0 = (float)(int)(int8_t)(e & 0xff);
                v1 = (float)(int)(int8_t)((e >> 8) & 0xff);
                v2 = (float)(int)(int8_t)((e >> 16) & 0xff);
                v3 = (float)(int)(int8_t)(e >> 24);
            } else {
                v0 = v2 = a.fill_re; v1 = v3 = a.fill_im;
            }
            if (two && (((uintptr_t)o & 15) == 0)) {
                bb_store4<NT>(o, bb_f4{v0, v1, v2, v3});
            } else {
                bb_store1<NT>(o, v0); bb_store1<NT>(o + 1, v1);
                if (two) { bb_store1<NT>(o + 2, v2); bb_store1<NT>(o + 3, v3); }
            }
        }
        __syncthreads();
    }
}


// Layouts whose INPUT rows are long but whose output position is strided
// (MKBF heaps: a (pol, chan) pair holds 256 consecutive times; GUPPI time-first:
// one time holds all (chan, pol)): stage the tile in input order -- coalesced
// dword loads, conflict-free dword LDS writes -- and do the permutation on the
// LDS READ side, where every lane picks the two 2-byte elements of its float4.
// The row pitch is an odd number of dwords.  (k_decode_i8_tiled, which places
// elements at their output position with 2-byte LDS writes and, for these two
// layouts, read 2-byte pieces from global memory, reached 3.1-3.7 TB/s here;
// profiles/r01i_exp_tiled.log.)
//   LAYOUT 1  MKBF: LDS row = (p, c), tt times long       (a.tcp = pitch in elements)
//   LAYOUT 2  GUPPI time-first: LDS row = one time, tc * npol elements
template <int LAYOUT, bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_i8_stage(bb_tiled_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t s_tile[];
    uint32_t *s32 = reinterpret_cast<uint32_t *>(s_tile);
    const uint32_t npol = a.npol, tc = a.tc, tt = a.tt, pe = a.tcp;
    const uint64_t rows_out = a.t_hi - a.t_lo;
    const uint64_t rowlen = (uint64_t)npol * a.nchan * 2;
    const uint64_t nwork = a.nframes * a.ntt * a.nct;
    const int lane = bb_lane(), wave = bb_wave();

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / ((uint64_t)a.ntt * a.nct);
        const uint32_t rem = (uint32_t)(work - f * a.ntt * a.nct);
        const uint32_t ti = rem / a.nct, ci = rem - ti * a.nct;
        const uint64_t t0 = a.t_lo + (uint64_t)ti * tt;
        const uint32_t nt_tile = (uint32_t)((a.t_hi - t0 < tt) ? a.t_hi - t0 : tt);
        const uint32_t c0 = ci * tc;
        const uint32_t nc_tile = (a.nchan - c0 < tc) ? a.nchan - c0 : tc;
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint16_t *in = reinterpret_cast<const uint16_t *>(a.buf + (valid ? so : 0));

        if (valid) {
            const uint32_t nrows_in = LAYOUT == 2 ? nt_tile : npol * nc_tile;
            const uint32_t run = LAYOUT == 2 ? nc_tile * npol : nt_tile;
            for (uint32_t r = wave; r < nrows_in; r += BB_WAVES_PER_BLOCK) {
                if (LAYOUT == 2) {
                    const uint16_t *row = in + (t0 + r) * a.st + (uint64_t)c0 * npol;    // (st = stored channels x npol)
                    if (((uintptr_t)row & 3) == 0) {
                        for (uint32_t i = lane * 2; i < run; i += 128) {
                            if (i + 1 < run) s32[(r * pe + i) >> 1] = *reinterpret_cast<const uint32_t *>(row + i);
                            else             s_tile[r * pe + i] = row[i];
                        }
                    } else {
                        for (uint32_t i = lane; i < run; i += 64) s_tile[r * pe + i] = row[i];
                    }
                } else {
                    const uint32_t p = r / nc_tile, c = r - p * nc_tile;
                    const uint16_t *rowbase = in + (uint64_t)p * a.sp + (uint64_t)(c0 + c) * a.sc;
                    // heaps hold 256 times (a.tb == 256): shifts, not divisions
                    for (uint32_t i = lane * 2; i < run; i += 128) {
                        const uint64_t t = t0 + i;
                        const uint16_t *ptr = rowbase + (t >> 8) * a.sh + (t & 255);
                        if (i + 1 < run && (t & 255) != 255 && ((uintptr_t)ptr & 3) == 0) {
                            s32[(r * pe + i) >> 1] = *reinterpret_cast<const uint32_t *>(ptr);
                        } else {
                            s_tile[r * pe + i] = ptr[0];
                            if (i + 1 < run) {
                                const uint64_t t1 = t + 1;
                                s_tile[r * pe + i + 1] = rowbase[(t1 >> 8) * a.sh + (t1 & 255)];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t pairs = (tc + 1) / 2;
        const uint32_t nrows = nt_tile * npol;
        float *obase = a.out + (f * rows_out + (t0 - a.t_lo)) * rowlen + (uint64_t)c0 * 2;
        for (uint32_t j = threadIdx.x; j < nrows * pairs; j += BB_BLOCK) {
            const uint32_t row = j / pairs;
            const uint32_t c = (j - row * pairs) * 2;
            if (c >= nc_tile) continue;
            float *o = obase + (uint64_t)row * a.nchan * 2 + (uint64_t)c * 2;
            const bool two = c + 1 < nc_tile;
            float v0, v1, v2 = 0.f, v3 = 0.f;
            if (valid) {
                const uint32_t tl = row / npol, p = row - tl * npol;
                const uint32_t i0 = LAYOUT == 2 ? tl * pe + c * npol + p : (p * nc_tile + c) * pe + tl;
                const uint32_t e0 = s_tile[i0];
                const uint32_t e1 = two ? s_tile[i0 + (LAYOUT == 2 ? npol : pe)] : 0u;
                v0 = (float)(int)(int8_t)(e0 & 0xff);
                v1 = (float)(int)(int8_t)(e0 >> 8);
                v2 = (float)(int)(int8_t)(e1 & 0xff);
                v3 = (float)(int)(int8_t)(e1 >> 8);
            } else {
                v0 = v2 = a.fill_re; v1 = v3 = a.fill_im;
            }
            if (two && (((uintptr_t)o & 15) == 0)) {
                bb_store4<NT>(o, bb_f4{v0, v1, v2, v3});
            } else {
                bb_store1<NT>(o, v0); bb_store1<NT>(o + 1, v1);
                if (two) { bb_store1<NT>(o + 2, v2); bb_store1<NT>(o + 3, v3); }
            }
        }
        __syncthreads();
    }
}
