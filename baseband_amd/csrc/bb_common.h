// Shared device/host helpers for libbbdecode (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "bbdecode.h"

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
