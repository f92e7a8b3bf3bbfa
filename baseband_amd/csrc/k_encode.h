// Encoders: float32 samples -> packed codes (the write-side twin of the
// decode kernels).  16x read-amplified for 2-bit
// data: bound by HBM reads.
//
// Replaces (reference, path:line): encode_1bit_base / encode_2bit_base /
// encode_4bit_base / encode_8bit (base/encoding.py:63-158) and the packing of
// vdif/payload.py:77-114, the sign/magnitude re-ordering of
// mark5b/payload.py:84-106, the integer encoders of gsb/payload.py:44-52 and
// dada/payload.py:17-18, and the track multiplexing of the Mark 4 encoders
// (mark4/payload.py:138-300).  Every threshold is evaluated with the same
// float32 operations NumPy performs, including its floor_divide algorithm.
#pragma once
#include "bb_common.h"

// numpy.floor_divide for float32 (npy_divmodf): fmod-based, so that quotients
// that round up to an integer in a plain division are still floored correctly.
__device__ __forceinline__ float bb_np_floor_divide(float a, float b)
{
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f) {
        if ((b < 0.0f) != (mod < 0.0f)) { mod += b; div -= 1.0f; }
    }
    if (div != 0.0f) {
        float fd = floorf(div);
        if (div - fd > 0.5f) fd += 1.0f;
        return fd;
    }
    return copysignf(0.0f, a / b);
}

template <int CODER, int BPS>
__device__ __forceinline__ uint32_t bb_encode_one(float x)
{
    if (BPS == 1) {
        if (CODER == BB_CODER_VDIF) return x >= 0.0f ? 1u : 0u;          // base/encoding.py:63-74
        return (__float_as_uint(x) >> 31);                                // np.signbit (mark5b)
    } else if (BPS == 2) {
        // base/encoding.py:77-102: clip to +-1.5 sigma, add 2 sigma, floor_divide by sigma
        const float sigma = 2.174564f;
        const float lo = (float)(-1.5 * 2.174564), hi = (float)(1.5 * 2.174564);
        float w = fminf(fmaxf(x, lo), hi);
        w = w + (float)(2 * 2.174564);
        const uint32_t c = (uint32_t)(int)bb_np_floor_divide(w, sigma);
        if (CODER == BB_CODER_MARK5B) return ((c & 1u) << 1) | (c >> 1);  // reorder [0, 2, 1, 3]
        return c;
    } else if (BPS == 4) {
        if (CODER == BB_CODER_VDIF) {                                      // base/encoding.py:105-128
            float w = x * 2.95f;
            w = w + 8.5f;
            w = fminf(fmaxf(w, 0.0f), 15.0f);
            return (uint32_t)(int)w;
        }
        float r = rintf(x);                                                // gsb/payload.py:44-48
        r = fminf(fmaxf(r, -8.0f), 7.0f);
        return (uint32_t)((int)r) & 0xfu;
    } else {
        if (CODER == BB_CODER_VDIF) {                                      // base/encoding.py:147-158
            float r = rintf(x * 35.5f + 127.5f);
            r = fminf(fmaxf(r, 0.0f), 255.0f);
            return (uint32_t)(int)r;
        }
        float r = rintf(x);                                                // dada/payload.py:17-18
        r = fminf(fmaxf(r, -128.0f), 127.0f);
        return (uint32_t)((int)r) & 0xffu;
    }
}

// One float4 (16 coalesced bytes) per lane -> 4*BPS bits of output.
template <int CODER, int BPS>
__global__ __launch_bounds__(BB_BLOCK)
void k_encode_flat(const float *in, uint64_t nquad, uint8_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * BB_BLOCK;
    const uint64_t nquad_pad = (nquad + 63) & ~63ull;           // keep waves whole for the shuffle
    for (uint64_t q = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x; q < nquad_pad; q += stride) {
        uint32_t bits = 0;
        if (q < nquad) {
            const bb_f4 v = __builtin_nontemporal_load(reinterpret_cast<const bb_f4 *>(in) + q);
            bits = bb_encode_one<CODER, BPS>(v.x)
                 | (bb_encode_one<CODER, BPS>(v.y) << BPS)
                 | (bb_encode_one<CODER, BPS>(v.z) << (2 * BPS))
                 | (bb_encode_one<CODER, BPS>(v.w) << (3 * BPS));
        }
        if (BPS == 8) {
            if (q < nquad) reinterpret_cast<uint32_t *>(out)[q] = bits;
        } else if (BPS == 4) {
            if (q < nquad) reinterpret_cast<uint16_t *>(out)[q] = (uint16_t)bits;
        } else if (BPS == 2) {
            if (q < nquad) out[q] = (uint8_t)bits;
        } else {
            // two lanes make one byte: even lane = low nibble
            const uint32_t other = (uint32_t)__shfl_xor((int)bits, 1);
            if (!(q & 1) && q < nquad) out[q >> 1] = (uint8_t)(bits | (other << 4));
        }
    }
}

struct bb_m4enc_args {
    const float *in;
    uint8_t *out;
    uint64_t nwords;
    uint32_t sign_bit[8];
    uint32_t mag_bit[8];
};

// Mark 4: NTRACK/8 lanes share one stream word; each encodes its four
// (sample, channel) values to 2-bit codes (sign = code >> 1, magnitude =
// code & 1), scatters them to their track bits, and the partial words are
// OR-reduced with xor shuffles.
template <int NTRACK>
__global__ __launch_bounds__(BB_BLOCK)
void k_encode_mark4(bb_m4enc_args a)
{
    typedef typename bb_m4_word<NTRACK>::type word_t;
    constexpr int LPW = NTRACK / 8;
    const int lane = bb_lane();
    const int sub = lane % LPW;
    uint32_t spack = 0, mpack = 0;
#pragma unroll
    for (int k = 0; k < LPW; ++k)
        if (sub == k) { spack = a.sign_bit[k]; mpack = a.mag_bit[k]; }
    const uint64_t nquad = a.nwords * LPW;
    const uint64_t nquad_pad = (nquad + 63) & ~63ull;
    const uint64_t stride = (uint64_t)gridDim.x * BB_BLOCK;
    for (uint64_t q = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x; q < nquad_pad; q += stride) {
        uint64_t part = 0;
        if (q < nquad) {
            const bb_f4 v = __builtin_nontemporal_load(reinterpret_cast<const bb_f4 *>(a.in) + q);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t c = bb_encode_one<BB_CODER_VDIF, 2>(vv[k]);
                part |= (uint64_t)(c >> 1) << ((spack >> (8 * k)) & 0xff);
                part |= (uint64_t)(c & 1u) << ((mpack >> (8 * k)) & 0xff);
            }
        }
#pragma unroll
        for (int d = 1; d < LPW; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)(part & 0xffffffffull), d);
            const uint32_t hi = NTRACK == 64 ? (uint32_t)__shfl_xor((int)(uint32_t)(part >> 32), d) : 0u;
            part |= ((uint64_t)hi << 32) | lo;
        }
        if (sub == 0 && q < nquad)
            reinterpret_cast<word_t *>(a.out)[q / LPW] = (word_t)part;
    }
}
