// Shared device/host helpers for libbbdecode (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "bbdecode.h"
#include "bbdecode_tune.h"

// BB_EXPERIMENTS (make EXPERIMENTS=1) builds libbbdecode_exp.so: the product
// kernels plus the measurement variants of rounds 1-2 and their knobs
// (include/bbdecode_exp.h).  The product library carries none of that.
#ifdef BB_EXPERIMENTS
#define BB_EXP 1
#include "bbdecode_exp.h"
#else
#define BB_EXP 0
#endif

#define BB_WAVE 64
#define BB_BLOCK 256            // 4 waves: one per SIMD of a CU
#define BB_WAVES_PER_BLOCK (BB_BLOCK / BB_WAVE)

typedef float  bb_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t bb_u4 __attribute__((ext_vector_type(4)));

// Level tables, indexed [coder][log2(bps)][code]; uploaded once per device by
// bb_init() from values computed on the host with IEEE float32 division so
// that they are bit-identical to the NumPy expressions of the reference
// (base/encoding.py:52-56,141-143).
__device__ float g_levels[3][4][256];

__device__ __forceinline__ int bb_lane() { return threadIdx.x & (BB_WAVE - 1); }
__device__ __forceinline__ int bb_wave() { return threadIdx.x / BB_WAVE; }

// Work order of the decode launches ("striping").  A launch's work items are
// numbered in output order; item w is NOT given to the w-th workgroup-step but
// dealt over 2^lw contiguous ranges ("stripes") of the launch: step w takes
// item (w % 2^lw) * stripe + w / 2^lw.  Workgroups that run side by side thus
// write into 2^lw places spread over the whole output of the launch instead of
// one moving window.  Why: on MI355X a decode-shaped store stream that stays
// inside one physical region of HBM runs at 5.4-5.65 TB/s, one that touches
// two different regions at the same time at 6.3-6.6 (profiles/r02b_exp_stripe.log,
// r02c_exp_stripe2.log: same kernel, same launch size, only the placement of
// the output differs); which regions an allocation lies in is the driver's
// choice, so the launch covers as much of its own output at once as it can.
// The map is a bijection on [0, n): the n % 2^lw items at the end keep their
// place.
struct bb_perm_t {
    uint64_t n;             // 2^lw * stripe: items that are dealt out (0 = identity)
    uint64_t stripe;        // items per stripe
    uint32_t lw;            // log2(number of stripes)
};

__device__ __forceinline__ uint64_t bb_perm(const bb_perm_t &p, uint64_t w)
{
    if (w >= p.n) return w;
    return (w & ((1ull << p.lw) - 1)) * p.stripe + (w >> p.lw);
}

// A source offset taken from an index is only followed when the whole unit it
// names lies inside the buffer: `lim` = buffer bytes - unit bytes + 1 (0 when
// the buffer is shorter than one unit), so one unsigned compare rejects
// negative offsets (-1 = missing frame) and offsets past the end alike; both
// decode as fill.  The reference never returns garbage for a short read either
// (EOFError, base/payload.py:135-136).
__device__ __forceinline__ bool bb_src_ok(int64_t so, uint64_t lim) { return (uint64_t)so < lim; }

// 16-byte store, optionally with the non-temporal hint (streamed output is
// never re-read by this library).
template <bool NT>
__device__ __forceinline__ void bb_store4(float *p, bb_f4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<bb_f4 *>(p));
    else    *reinterpret_cast<bb_f4 *>(p) = v;
}

template <bool NT>
__device__ __forceinline__ void bb_store1(float *p, float v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else    *p = v;
}
