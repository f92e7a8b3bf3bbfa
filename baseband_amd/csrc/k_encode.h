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
__host__ __device__ __forceinline__ float bb_np_floor_divide_hd(float a, float b)
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

// The 2-bit code is a monotone step function of the input, so it is fully
// described by the three smallest floats at which it reaches 1, 2 and 3.
// bb_init() finds them by bisection over the float32 bit patterns with the
// exact reference arithmetic (bb_encode2_reference below, compiled for the
// host), so the default path is three compares per sample.  DIRECT evaluates
// the reference arithmetic on the device instead (BB_TUNE_ENCODE_DIRECT; the
// GPU tests check both paths against each other over all 2^32 inputs).
__device__ float g_enc2_thr[3];

__host__ __device__ __forceinline__ float bb_np_floor_divide_hd(float a, float b);

__host__ __device__ inline uint32_t bb_encode2_reference(float x)
{
    // base/encoding.py:77-102: clip to +-1.5 sigma, add 2 sigma, floor_divide by sigma
    const float sigma = 2.174564f;
    const float lo = (float)(-1.5 * 2.174564), hi = (float)(1.5 * 2.174564);
    float w = x < lo ? lo : x;              // np.clip = minimum(maximum(x, lo), hi)
    w = w > hi ? hi : w;
    w = w + (float)(2 * 2.174564);
    return (uint32_t)(int)bb_np_floor_divide_hd(w, sigma);
}

template <int CODER, int BPS, bool DIRECT = false>
__device__ __forceinline__ uint32_t bb_encode_one(float x)
{
    if (BPS == 1) {
        if (CODER == BB_CODER_VDIF) return x >= 0.0f ? 1u : 0u;          // base/encoding.py:63-74
        return (__float_as_uint(x) >> 31);                                // np.signbit (mark5b)
    } else if (BPS == 2) {
        uint32_t c;
        if (DIRECT) c = bb_encode2_reference(x);
        else c = (uint32_t)(x >= g_enc2_thr[0]) + (uint32_t)(x >= g_enc2_thr[1])
               + (uint32_t)(x >= g_enc2_thr[2]);
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

template <int CODER, int BPS, bool DIRECT>
__device__ __forceinline__ uint32_t bb_encode_quad(const bb_f4 v)
{
    return bb_encode_one<CODER, BPS, DIRECT>(v.x)
         | (bb_encode_one<CODER, BPS, DIRECT>(v.y) << BPS)
         | (bb_encode_one<CODER, BPS, DIRECT>(v.z) << (2 * BPS))
         | (bb_encode_one<CODER, BPS, DIRECT>(v.w) << (3 * BPS));
}

// Quad q (4 floats = 16 input bytes) -> 4*BPS bits of output.  A wave takes
// runs of 256 quads (4 KiB in): four coalesced float4 loads per lane are in
// flight before the first compare; for 2-bit data the four result bytes of a
// lane are transposed across the wave (ds_bpermute) so that the run leaves as
// ONE coalesced 256-byte store.  Quads past the last whole run go through the
// per-quad tail at the end.
// RUNS: runs a wave takes per step (its loads are all in flight before the
// first compare: 4 x RUNS float4 per lane).
template <int CODER, int BPS, bool DIRECT, int RUNS = 1>
__global__ __launch_bounds__(BB_BLOCK)
void k_encode_flat(const float *in, uint64_t nquad, uint8_t *out, bb_perm_t perm)
{
    const int lane = bb_lane();
    const bb_f4 *in4 = reinterpret_cast<const bb_f4 *>(in);
    const uint64_t nrun = nquad >> 8;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x) >> 6;
    const uint64_t nwave = ((uint64_t)gridDim.x * BB_BLOCK) >> 6;
    for (uint64_t rr = wave0 * RUNS; rr < nrun; rr += nwave * RUNS) {
      bb_f4 vv[RUNS][4];
#pragma unroll
      for (int h = 0; h < RUNS; ++h) {
          const uint64_t q0 = (bb_perm(perm, rr + h) << 8) + lane;
#pragma unroll
          for (int j = 0; j < 4; ++j)
              vv[h][j] = (rr + h < nrun) ? __builtin_nontemporal_load(in4 + q0 + 64 * j) : bb_f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int h = 0; h < RUNS; ++h) {
        if (rr + h >= nrun) break;                              // (wave-uniform)
        const uint64_t r = bb_perm(perm, rr + h);
        const uint64_t q0 = (r << 8) + lane;
        uint32_t bits[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bits[j] = bb_encode_quad<CODER, BPS, DIRECT>(vv[h][j]);
        if (BPS == 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) reinterpret_cast<uint32_t *>(out)[q0 + 64 * j] = bits[j];
        } else if (BPS == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) reinterpret_cast<uint16_t *>(out)[q0 + 64 * j] = (uint16_t)bits[j];
        } else if (BPS == 2) {
            // byte of (load j, lane l) sits at run offset 64 j + l; output dword d
            // = bytes of load d/16, lanes 4 (d%16) .. +3
            const uint32_t mine = bits[0] | (bits[1] << 8) | (bits[2] << 16) | (bits[3] << 24);
            const int sel = 8 * (lane >> 4), src = 4 * (lane & 15);
            uint32_t word = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                word |= (((uint32_t)__shfl((int)mine, src + k) >> sel) & 0xffu) << (8 * k);
            reinterpret_cast<uint32_t *>(out)[(r << 6) + lane] = word;
        } else {
            // nibble per quad: even lane = low nibble of the byte
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t other = (uint32_t)__shfl_xor((int)bits[j], 1);
                if (!(lane & 1)) out[(q0 + 64 * j) >> 1] = (uint8_t)(bits[j] | (other << 4));
            }
        }
      }
    }
    // tail: fewer than 256 quads, first workgroup only (whole waves for the shuffle)
    if (blockIdx.x == 0) {
        const uint64_t q = (nrun << 8) + threadIdx.x;
        const uint32_t bits = q < nquad ? bb_encode_quad<CODER, BPS, DIRECT>(in4[q]) : 0u;
        if (BPS == 8) {
            if (q < nquad) reinterpret_cast<uint32_t *>(out)[q] = bits;
        } else if (BPS == 4) {
            if (q < nquad) reinterpret_cast<uint16_t *>(out)[q] = (uint16_t)bits;
        } else if (BPS == 2) {
            if (q < nquad) out[q] = (uint8_t)bits;
        } else {
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
template <int NTRACK, bool DIRECT>
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
                const uint32_t c = bb_encode_one<BB_CODER_VDIF, 2, DIRECT>(vv[k]);
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
