// int8 (re, im) pairs -> complex64 with an axis permutation: the fast form.
//
// Replaces (reference, path:line) what k_tiled.h replaces --
//   GUPPI channels-first  words.reshape(nchan, -1, npol) -> (time, pol, chan)   (guppi/payload.py:90-96)
//   GUPPI time-first      .reshape(-1, nchan, npol).transpose(0, 2, 1)          (guppi/payload.py:97-102)
//   MKBF heaps            np.moveaxis(words["heaps"], -1, 1)                    (dada/payload.py:76-79)
//   and the int8 -> float32 casts (guppi/payload.py:13-14, dada/payload.py:13-14)
// -- for the common geometry: payloads whose input runs are 16-byte aligned
// (bbdecode.hip checks; everything else stays with k_tiled.h).
//
// What is different from k_tiled.h (0.57-0.61 of the HBM peak there):
//  * 16 bytes per lane along the contiguous INPUT axis (a tile's 16 KiB arrive
//    with four wave-wide 4 KiB loads per workgroup instead of sixteen 1 KiB ones);
//  * the LDS image is made of DWORDS, never 2-byte pieces: an input dword is two
//    elements that are neighbours along the input axis; it is written as it is
//    (conflict-free ds_write_b32, row pitch odd) and the store phase reads the two
//    dwords that hold its (chan, chan + 1) pair and takes the half it needs;
//  * persistent grid, software pipelined: the loads of a workgroup's NEXT tile
//    are in flight (in registers) while the current tile is being stored, as in
//    k_decode_flat_aln;
//  * work dealt over 16 stripes of the launch (bb_perm_t).
//
// A tile (TC = 64) is ROWS (128 or 64) output rows (of up to 64 channels = 512 bytes each)
// x 64 channels; lanes 0-31 of a store instruction write one output row, lanes
// 32-63 the next, so a wave stores 1 KiB contiguous when the tile spans all
// channels.
//   LAYOUT 0  GUPPI (chan, time, pol): input row = a channel, elements i = t * npol + p;
//             LDS D[i / 2][chan]; output row = i
//   LAYOUT 1  MKBF (heap, pol, chan, 256 times): input row = (pol, chan), 128 / npol times;
//             LDS D[pol][t / 2][chan]; output row = t * npol + pol
//   LAYOUT 2  GUPPI (time, chan, pol), npol = 2: input row = a time (64 channels x 2 pol);
//             LDS D[t][chan] (a dword = both pols of one channel); output row = t * 2 + pol
//
// A reader `subset` that keeps some channels (any list: a.cmap) and / or one of
// two polarisations (a.nps stored, a.npol decoded from a.pf on) is folded in
// (the reference decodes whole blocks and indexes afterwards:
// base/base.py:706-717 after guppi/payload.py:90-102, dada/payload.py:76-79):
// channel c of a tile is loaded from stored channel cmap[c0 + c] -- rows of the
// other channels are never read in layouts 0 and 1; in layout 2, whose rows hold
// all channels of a time, a mapped tile loads one dword (both pols of a channel)
// per channel instead of 16-byte pieces.  A dropped polarisation: layout 1
// loads only the kept (pol, chan) rows; layouts 0 and 2 hold both pols in every
// LDS dword, so the image is the same and the store phase takes output row r
// from element i = r * (nps / npol) + pf: only the kept half is written.
#pragma once
#include "k_tiled.h"

// Channels per tile TC = 64, or 32 / 16 / 8 for narrow outputs (few channels
// stored, or few kept by a selection): the tile keeps its 16 KiB of input, so a
// narrow tile is RT = ROWS * 64 / TC output rows long and no load or store lane
// idles on channels that are not there.  LDS row pitch TC + 1 dwords (odd).

// (launch bounds: at least 6 waves per SIMD = 6 workgroups per CU; with the
// store loop fully unrolled the compiler took 123-138 VGPRs = 3-4 workgroups per
// CU, and a store-bound kernel whose waves all wait at the same barrier needs
// more of them in flight: profiles/r02e_kernels.csv)
// (MAPPED: a channel list -- its own instantiation of the body, so that the plain
// path's loads are not tied to the look-ups' waits where the two would merge)
template <int LAYOUT, bool NT, int ROWS, int TC, bool MAPPED>
__device__ __forceinline__ void bb_xpose_body(const bb_tiled_args &a, uint32_t *s_d)
{
    static_assert(TC == 64 || TC == 32 || TC == 16 || TC == 8, "channels per tile");
    constexpr int RT = ROWS * 64 / TC;          // elements of a (time, pol) run per tile
    constexpr int PITCH = TC + 1;
    constexpr int NLOAD = ROWS / 32;            // 16-byte loads per thread and tile
    constexpr uint32_t LPR = RT / 8;            // LAYOUT 0: pieces per channel row (power of two)
    constexpr uint32_t PPT = TC / 4;            // LAYOUT 2: pieces of 4 channels per time
    constexpr uint32_t PR = TC / 2;             // store phase: channel pairs per row ...
    constexpr uint32_t RPI = BB_BLOCK / PR;     // ... and rows per pass of the workgroup
    const uint32_t npol = a.npol, nps = a.nps;
    // output rows per LDS image: ROWS, or half of it when one of two pols is dropped (layouts 0, 2)
    const uint32_t estep = LAYOUT == 1 ? 1u : nps / npol;
    const uint32_t rpt = RT / estep;
    const uint64_t rows_out = (a.t_hi - a.t_lo) * npol;         // output rows per frame
    const uint64_t rowlen = (uint64_t)a.nchan * 2;              // floats per output row
    // tile grid of a frame: a.ntt tiles along the output rows, a.nct along channels
    const uint64_t per_frame = (uint64_t)a.ntt * a.nct;
    const uint64_t nwork = a.nframes * per_frame;
    const uint32_t tid = threadIdx.x;
    // times per tile
    const uint32_t tt = LAYOUT == 0 ? 0u : LAYOUT == 1 ? RT / npol : RT / nps;
    auto cm = [&](uint32_t c) -> uint32_t { return MAPPED ? (uint32_t)a.cmap[c] : c; };

    bb_u4 nxt[NLOAD];
    bool nxt_keep[NLOAD];
    bool nxt_valid = false;

    // which piece of the tile this thread loads in round k (k = 0..3)
    auto issue = [&](uint64_t step, bb_u4 (&w)[NLOAD], bool (&keep)[NLOAD], bool &valid) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / per_frame;
        const uint32_t rem = (uint32_t)(work - f * per_frame);
        const uint32_t ti = rem / a.nct, ci = rem - ti * a.nct;
        const uint32_t c0 = ci * TC;
        const uint32_t ncv = (a.nchan - c0 < TC) ? a.nchan - c0 : TC;
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        valid = bb_src_ok(so, a.src_lim);
        const uint16_t *in = reinterpret_cast<const uint16_t *>(a.buf + (valid ? so : 0));
        // Addresses first (a mapped channel costs a look-up, and the wait for it also
        // waited for the 16-byte load issued just before: one load in flight per thread
        // instead of NLOAD; round 5), then the loads back to back.
        const uint16_t *ptrs[NLOAD];
        bool wants[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const uint32_t g = (uint32_t)k * BB_BLOCK + tid;        // 16-byte piece of the tile
            const uint16_t *ptr = in;
            bool want = valid;
            if (LAYOUT == 0) {
                // TC channel rows x RT / 8 pieces of 8 elements
                const uint32_t c = g / LPR, piece = g % LPR;
                const uint64_t i0 = a.t_lo * nps + (uint64_t)ti * RT;
                const uint64_t i = i0 + piece * 8;
                want = want && c < ncv && i < a.t_hi * nps;
                if (want) ptr = in + (uint64_t)cm(c0 + c) * a.sc + i;
            } else if (LAYOUT == 1) {
                // npol * TC (pol, chan) rows x (RT / npol / 8) pieces of 8 times
                const uint32_t ppr = tt >> 3;                       // pieces per row
                const uint32_t row = g / ppr, piece = g - row * ppr;
                const uint32_t p = row / TC, c = row % TC;
                const uint64_t t = a.t_lo + (uint64_t)ti * tt + piece * 8;
                want = want && c < ncv && t < a.t_hi;
                if (want) ptr = in + (t >> 8) * a.sh + (t & 255) + (uint64_t)(a.pf + p) * a.sp + (uint64_t)cm(c0 + c) * a.sc;
            } else {
                // RT / 2 times x TC / 4 pieces of 4 channels (both pols)
                const uint32_t tl = g / PPT, piece = g % PPT;
                const uint64_t t = a.t_lo + (uint64_t)ti * tt + tl;
                want = want && t < a.t_hi && piece * 4 < ncv;
                ptr = in + t * a.st + (uint64_t)(c0 + piece * 4) * 2;     // (st = stored channels x 2 pol)
            }
            ptrs[k] = ptr;
            wants[k] = want;
        }
        if (LAYOUT == 2 && MAPPED) {
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                // mapped channels are not neighbours: one dword (both pols) per channel
                const uint32_t g = (uint32_t)k * BB_BLOCK + tid;
                const uint32_t tl = g / PPT, piece = g % PPT;
                const uint64_t t = a.t_lo + (uint64_t)ti * tt + tl;
                bb_u4 v = {0u, 0u, 0u, 0u};
                if (wants[k]) {
                    const uint16_t *row = in + t * a.st;
                    const uint32_t cb = piece * 4;
                    v.x = *reinterpret_cast<const uint32_t *>(row + (uint64_t)cm(c0 + cb) * 2);
                    if (cb + 1 < ncv) v.y = *reinterpret_cast<const uint32_t *>(row + (uint64_t)cm(c0 + cb + 1) * 2);
                    if (cb + 2 < ncv) v.z = *reinterpret_cast<const uint32_t *>(row + (uint64_t)cm(c0 + cb + 2) * 2);
                    if (cb + 3 < ncv) v.w = *reinterpret_cast<const uint32_t *>(row + (uint64_t)cm(c0 + cb + 3) * 2);
                }
                w[k] = v;
                keep[k] = true;
            }
            return;
        }
        // (pieces outside the tile read the first bytes of the buffer and are zeroed
        // afterwards: loads without a branch around them stay in flight together ...
#pragma unroll
        for (int k = 0; k < NLOAD; ++k)
            w[k] = *reinterpret_cast<const bb_u4 *>(wants[k] ? ptrs[k] : reinterpret_cast<const uint16_t *>(a.buf));
        // (... when they are written to LDS, not here: a use of the values at this point
        // would wait for the loads that are meant to fly during the stores of the tile before)
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) keep[k] = wants[k];
    };

    uint64_t step = blockIdx.x;
    if (step < nwork) issue(step, nxt, nxt_keep, nxt_valid);
    for (; step < nwork; step += gridDim.x) {
        // registers -> LDS image of this tile
        const bool valid = nxt_valid;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const uint32_t g = (uint32_t)k * BB_BLOCK + tid;
            if (!nxt_keep[k]) nxt[k] = bb_u4{0u, 0u, 0u, 0u};
            uint32_t base;
            if (LAYOUT == 0) {
                const uint32_t c = g / LPR, piece = g % LPR;
                base = (piece * 4) * PITCH + c;               // D[piece*4 + j][c]
                s_d[base] = nxt[k].x;
                s_d[base + PITCH] = nxt[k].y;
                s_d[base + 2 * PITCH] = nxt[k].z;
                s_d[base + 3 * PITCH] = nxt[k].w;
            } else if (LAYOUT == 1) {
                const uint32_t ppr = tt >> 3;
                const uint32_t row = g / ppr, piece = g - row * ppr;
                const uint32_t p = row / TC, c = row % TC;
                base = (p * (tt >> 1) + piece * 4) * PITCH + c;   // D[p][piece*4 + j][c]
                s_d[base] = nxt[k].x;
                s_d[base + PITCH] = nxt[k].y;
                s_d[base + 2 * PITCH] = nxt[k].z;
                s_d[base + 3 * PITCH] = nxt[k].w;
            } else {
                const uint32_t tl = g / PPT, piece = g % PPT;
                base = tl * PITCH + piece * 4;                // D[t][piece*4 + j]
                s_d[base] = nxt[k].x;
                s_d[base + 1] = nxt[k].y;
                s_d[base + 2] = nxt[k].z;
                s_d[base + 3] = nxt[k].w;
            }
        }
        __syncthreads();
        const uint64_t next = step + gridDim.x;
        if (next < nwork) issue(next, nxt, nxt_keep, nxt_valid);

        // LDS -> global
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / per_frame;
        const uint32_t rem = (uint32_t)(work - f * per_frame);
        const uint32_t ti = rem / a.nct, ci = rem - ti * a.nct;
        const uint32_t c0 = ci * TC;
        const uint32_t ncv = (a.nchan - c0 < TC) ? a.nchan - c0 : TC;
        const uint64_t row0 = (uint64_t)ti * rpt;             // first output row of the tile in its frame
        float *obase = a.out + (f * rows_out + row0) * rowlen + (uint64_t)c0 * 2;
        const uint64_t rows_left = rows_out - row0;
        const uint32_t cp = tid % PR;                               // channel pair
        const uint32_t rsub = tid / PR;                             // row within a pass
        const bb_f4 fillv = {a.fill_re, a.fill_im, a.fill_re, a.fill_im};
#pragma unroll 4
        for (int q = 0; q < ROWS / 8; ++q) {
            const uint32_t r = (uint32_t)q * RPI + rsub;              // output row of the tile
            uint32_t idx, half;
            if (LAYOUT == 1) {
                const uint32_t tl = npol == 2 ? r >> 1 : r, p = npol == 2 ? r & 1 : 0;
                idx = (p * (tt >> 1) + (tl >> 1)) * PITCH; half = tl & 1;
            } else {
                // element i of the (time, pol) run: row r as it is, or -- one of two
                // pols dropped -- the kept pol of time r
                const uint32_t i = r * estep + a.pf;
                idx = (i >> 1) * PITCH; half = i & 1;
            }
            if (r >= rpt) continue;
            const uint32_t x = s_d[idx + 2 * cp], y = s_d[idx + 2 * cp + 1];
            if (r >= rows_left || 2 * cp >= ncv) continue;
            const uint32_t e0 = half ? x >> 16 : x & 0xffffu;
            const uint32_t e1 = half ? y >> 16 : y & 0xffffu;
            bb_f4 v;
            if (valid) {
                v.x = (float)(int)(int8_t)(e0 & 0xff);
                v.y = (float)(int)(int8_t)(e0 >> 8);
                v.z = (float)(int)(int8_t)(e1 & 0xff);
                v.w = (float)(int)(int8_t)(e1 >> 8);
            } else {
                v = fillv;
            }
            float *o = obase + (uint64_t)r * rowlen + 4 * cp;
            if (2 * cp + 1 < ncv) bb_store4<NT>(o, v);
            else { bb_store1<NT>(o, v.x); bb_store1<NT>(o + 1, v.y); }
        }
        __syncthreads();
    }
}

template <int LAYOUT, bool NT, int ROWS, int TC>
__global__ __launch_bounds__(BB_BLOCK, 6)
void k_decode_i8_xpose(bb_tiled_args a)
{
    __shared__ uint32_t s_d[(ROWS * 64 / TC / 2) * (TC + 1)];
    if (a.cmap) bb_xpose_body<LAYOUT, NT, ROWS, TC, true>(a, s_d);
    else bb_xpose_body<LAYOUT, NT, ROWS, TC, false>(a, s_d);
}
