// A channel LIST of time-first int8 blocks (GUPPI (time, chan, pol), two
// polarisations stored): the reader's `subset` folded into the decode
// (base/base.py:706-717 after guppi/payload.py:97-102; the reference decodes
// whole blocks and indexes afterwards).
//
// Round 3 ran these through k_decode_i8_xpose with ONE dword load per mapped
// channel -- every row of the block is read anyway (all channels of a time
// share its cache lines), but through 4-byte gathers: 4.5-5.0 TB/s of bytes
// moved.  Here a tile is T consecutive times = T x (stored channels x 4) bytes,
// which are CONTIGUOUS in the block: the workgroup brings them in with
// direct-to-LDS 16-byte loads (global_load_lds_dwordx4, 1 KiB per wave
// instruction), picks the kept channels out of LDS and writes the tile's
// output, which is contiguous too (time, pol, kept channel): float4 stores,
// 1 KiB per wave instruction.  One tile per workgroup, striped work order.
#pragma once
#include "k_tiled.h"

#define BB_TFPICK_STAGE 16384u      // bytes of input per tile
#define BB_TFPICK_MAXSEL 2048u      // kept channels at most (their map sits in LDS)

struct bb_tfpick_args {
    const uint8_t *buf;
    float *out;
    const int32_t *cmap;            // [nsel] stored channel of kept channel j
    uint64_t nframes;
    uint64_t t_lo, t_hi;
    int64_t  src0, src_stride;
    uint64_t src_lim;
    uint32_t rb;                    // bytes per time: stored channels x 2 pols x 2
    uint32_t nsel;                  // kept channels (even)
    uint32_t npd, pf;               // polarisations decoded (1 or 2), first one
    uint32_t tt;                    // times per tile
    uint32_t ntt;                   // tiles per frame
    uint32_t magic_ppt, magic_half; // floor(2^32 / d) + 1 for d = float4 per time, pairs per (time, pol)
    float fill_re, fill_im;
    bb_perm_t perm;
};

template <bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_i8_tf_pick(bb_tfpick_args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[BB_TFPICK_STAGE];
    __shared__ int32_t s_map[BB_TFPICK_MAXSEL];
    for (uint32_t j = threadIdx.x; j < a.nsel; j += BB_BLOCK) s_map[j] = a.cmap[j];
    const uint32_t half = a.nsel >> 1;                          // channel pairs per (time, pol)
    const uint32_t ppt = half * a.npd;                          // float4 per time
    const uint64_t rows = a.t_hi - a.t_lo;
    const uint64_t nwork = a.nframes * a.ntt;
    const bb_f4 fillv = {a.fill_re, a.fill_im, a.fill_re, a.fill_im};
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    const int lane = bb_lane();
    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / a.ntt;
        const uint32_t ti = (uint32_t)(work - f * a.ntt);
        const uint64_t t0 = a.t_lo + (uint64_t)ti * a.tt;
        const uint32_t nt = (uint32_t)(a.t_hi - t0 < a.tt ? a.t_hi - t0 : a.tt);
        const int64_t so = a.src0 + (int64_t)f * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint32_t nb = nt * a.rb;                          // bytes of the tile (a multiple of 16)
        __syncthreads();                                        // the map is there / the stage is free again
        if (valid) {
            const uint8_t *in = a.buf + (uint64_t)so + t0 * a.rb;
            for (uint32_t p0 = (uint32_t)wave * 1024; p0 < nb; p0 += BB_WAVES_PER_BLOCK * 1024) {
                if (p0 + 16 * (uint32_t)lane < nb)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + p0 + 16 * lane),
                                                     (__attribute__((address_space(3))) void *)(s_in + p0), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const uint32_t nf4 = nt * ppt;
        float *obase = a.out + ((f * rows + (t0 - a.t_lo)) * a.npd) * (uint64_t)a.nsel * 2;
        const uint32_t *d = reinterpret_cast<const uint32_t *>(s_in);
        const uint32_t rdw = a.rb >> 2;
        for (uint32_t o = threadIdx.x; o < nf4; o += BB_BLOCK) {
            bb_f4 v = fillv;
            if (valid) {
                const uint32_t t = ppt == 1 ? o : __umulhi(o, a.magic_ppt);      // (the magic of 1 does not fit 32 bits)
                const uint32_t r = o - t * ppt;
                const uint32_t p = a.npd == 2 ? (half == 1 ? r : __umulhi(r, a.magic_half)) : 0u;
                const uint32_t jp = r - p * half;
                const uint32_t x = d[t * rdw + (uint32_t)s_map[2 * jp]];
                const uint32_t y = d[t * rdw + (uint32_t)s_map[2 * jp + 1]];
                const uint32_t hs = (a.pf + p) ? 16u : 0u;
                const uint32_t e0 = (x >> hs) & 0xffffu, e1 = (y >> hs) & 0xffffu;
                v.x = (float)(int)(int8_t)(e0 & 0xff);
                v.y = (float)(int)(int8_t)(e0 >> 8);
                v.z = (float)(int)(int8_t)(e1 & 0xff);
                v.w = (float)(int)(int8_t)(e1 >> 8);
            }
            bb_store4<NT>(obase + 4 * (uint64_t)o, v);
        }
    }
}
