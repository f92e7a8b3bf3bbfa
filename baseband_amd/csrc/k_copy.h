// Strided frame copy: `nframes` runs of n bytes at src0 + f * src_stride ->
// one contiguous output.  The "decode" of samples that are float32 already --
// DADA NBIT 32, an EXTENSION of this package: the reference knows NBIT 8 only
// (dada/payload.py:40-41 raises KeyError(32)), so there is nothing to be
// bit-exact against except the file's own bytes (byte identity is the test).
//
// Shape (gfx950): a work item is 16 KiB of one frame (1024 16-byte pieces): a
// workgroup of 256 lanes issues its four non-temporal 16-byte loads per lane
// before the first store, then four non-temporal stores -- consecutive lanes on
// consecutive pieces, 4 KiB contiguous per wave instruction pair; one item per
// workgroup-step on a large grid, in the striped work order of the decode
// launches (bb_perm_t): reads are half of the traffic here, and many short
// workgroups overlap them with the stores best, as for the 8-bit kernels.
// Runs that are not 16-byte aligned (source, destination or length) are moved
// dword by dword.
#pragma once
#include "bb_common.h"

struct bb_copy_args {
    const uint8_t *buf;
    uint8_t *out;
    uint64_t nframes;
    uint64_t n;             // bytes per frame (multiple of 4)
    uint64_t nseg;          // work items per frame
    int64_t  src0, src_stride;
    bb_perm_t perm;
};

#define BB_COPY_ITEM 16384u

// NL: 16-byte loads a lane has in flight before its first store (work item = NL x
// 4 KiB); NTL: non-temporal loads.  The product builds <.., 4, true>; the
// experiment build the others (tools/experiments/exp_copy.py, profiles/r04k_exp_copy.log).
template <bool NT, bool V16, int NL = 4, bool NTL = true>
__global__ __launch_bounds__(BB_BLOCK)
void k_copy_frames(bb_copy_args a)
{
    constexpr uint32_t ITEM = (uint32_t)NL * BB_BLOCK * 16;
    const uint64_t nwork = a.nframes * a.nseg;
    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = a.nseg == 1 ? work : work / a.nseg;
        const uint64_t seg = work - f * a.nseg;
        const uint64_t b0 = seg * ITEM;
        const uint32_t nb = (uint32_t)(a.n - b0 < ITEM ? a.n - b0 : ITEM);
        const uint8_t *src = a.buf + (uint64_t)(a.src0 + (int64_t)f * a.src_stride) + b0;
        uint8_t *dst = a.out + f * a.n + b0;
        if (V16) {
            bb_u4 v[NL];
            const uint32_t np = nb >> 4;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const uint32_t p = (uint32_t)k * BB_BLOCK + threadIdx.x;
                if (p < np) v[k] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const bb_u4 *>(src) + p)
                                       : reinterpret_cast<const bb_u4 *>(src)[p];
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const uint32_t p = (uint32_t)k * BB_BLOCK + threadIdx.x;
                if (p < np) {
                    if (NT) __builtin_nontemporal_store(v[k], reinterpret_cast<bb_u4 *>(dst) + p);
                    else    reinterpret_cast<bb_u4 *>(dst)[p] = v[k];
                }
            }
            // (n is a multiple of 16 in this instantiation)
        } else {
            const uint32_t nd = nb >> 2;
#pragma nounroll
            for (uint32_t p = threadIdx.x; p < nd; p += BB_BLOCK)
                reinterpret_cast<uint32_t *>(dst)[p] = reinterpret_cast<const uint32_t *>(src)[p];
        }
    }
}
