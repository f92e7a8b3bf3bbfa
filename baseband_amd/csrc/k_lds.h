// The byte-table decode with 16-BYTE input loads staged through LDS (VERDICT r2
// next 4) -- since round 4 by DIRECT-TO-LDS loads (GL: global_load_lds_dwordx4,
// no VGPR round trip) -- the product's kernel for contiguous 2-bit output (the
// headline kernel, 6 tiles per wave), 4-bit output (4 tiles) and int8 output
// (k_decode_flat_lds<8,..,INT8,..,GL>, 4 tiles); 1-bit samples stay with
// k_decode_flat_lut (k_lut.h).  Measurements: round 3, register staging against
// k_decode_flat_lut's dword-per-lane loads handed out by ds_bpermute: +1.0-1.5 %
// at the headline size, +1.3-5.4 % at 2^16-2^18 frames, 4-bit -3..-12 %
// (profiles/r03j_exp_lds.log); round 4, direct-to-LDS against register staging:
// 2-bit +2.4-2.8 % with an index on three boxes (r04d_exp_glds3_box*.log), 4-bit
// +2.6-8.5 % against k_decode_flat_lut (r04r_exp_glds5_box*.log), int8 +1.2-4.7 %
// against the plain kernel (r04c, r04d); 1-bit -18 %.
//
// A lane that loads 16 contiguous bytes holds the codes of 64 (2-bit) samples
// = 256 B of output, but the store pattern that HBM wants is 1 KiB contiguous
// per wave instruction -- lane l writes float4 number 64 p + l -- so with wide
// loads the bits a lane needs sit in ANOTHER lane's register, in a component
// that differs from lane to lane: ds_bpermute (one VGPR per source lane) cannot
// hand them out.  Hence LDS: a wave writes its (up to) 1 KiB + 256 B of packed
// bytes with ds_write_b128, then every store pass reads its byte (ds_read_u8:
// 64 consecutive bytes per wave, conflict free), the byte's float4 from the
// table (ds_read_b128) and stores.  Misalignment against 256-byte blocks is an
// LDS address offset here, not a second shuffle.  Payload edges: a 16-byte
// piece that is not entirely inside the payload is loaded dword by dword, only
// the dwords inside (nothing outside the payload is read, as in k_lut.h).
// Payloads at odd byte addresses (files repaired by the byte-granular search)
// are staged byte by byte: rare, correctness only.
//
// One work item per workgroup-step (2 waves x tpw tiles), no register prefetch
// -- k_decode_flat_lut runs one item per workgroup at its default grid too.
#pragma once
#include "k_flat.h"

// 8-BIT samples (VDIF 8-bit, DADA / GSB / GUPPI-real int8; round 3): the same
// staging with nothing to look up per bit field -- a lane's float4 is ONE staged
// dword (ds_read_b32, consecutive lanes consecutive dwords), converted by cast
// (LV = BB_LV_INT8) or through the 256-entry level table (BB_LV_LDS).  These
// payloads read 20 % of their traffic, not 6 %: 16-byte loads matter more.
// TAG only names an instantiation: the output arena probes new memory with
// launches of TAG 1, so that a profile's per-kernel statistics of the decode
// launches proper (TAG 0) are not averaged with the probes' short ones.
// GL: whole 16-byte pieces by global_load_lds_dwordx4 straight into the stage, no VGPR
// round trip (the product's form since round 4).  AUX: cache
// policy bits of those loads (experiment build: 1 = sc0, 2 = nt, 3 = both).
template <int BPS, bool NT, int NW, int MAXT, int LV = BB_LV_REG, int TAG = 0, bool GL = false, int AUX = 0>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_flat_lds(bb_flat_args a)
{
    static_assert(BPS == 1 || BPS == 2 || BPS == 4 || BPS == 8, "1-, 2-, 4- or 8-bit samples");
    constexpr int EPT = 2048 / BPS;
    constexpr int PASSES = 8 / BPS;
    constexpr int FPB = BPS == 1 ? 2 : 1;
    constexpr uint32_t CMASK = (1u << BPS) - 1;
    constexpr int NPIECE = (MAXT + 1) * 16;                 // 16-byte pieces a wave stages at most
    constexpr int NLOAD = (NPIECE + BB_WAVE - 1) / BB_WAVE;
    __shared__ bb_f4 s_lut[BPS >= 4 ? 1 : 256 * FPB];
    __shared__ float2 s_lut2[BPS == 4 ? 256 : 1];
    __shared__ float s_tab8[(BPS == 8 && LV == BB_LV_LDS) ? 256 : 1];
    __shared__ bb_u4 s_stage[NW][NPIECE];
    if (BPS == 8) {
        if (LV == BB_LV_LDS)
            for (int i = threadIdx.x; i < 256; i += NW * BB_WAVE) s_tab8[i] = a.tab[i];
    } else if (BPS == 4) {
        for (int i = threadIdx.x; i < 256; i += NW * BB_WAVE)
            s_lut2[i] = float2{a.tab[i & 15], a.tab[i >> 4]};
    } else {
        for (int i = threadIdx.x; i < 256 * FPB; i += NW * BB_WAVE) {
            const uint32_t b = (uint32_t)i / FPB, h = (uint32_t)i % FPB;
            const uint32_t q = b >> (4 * BPS * h);
            s_lut[i] = bb_f4{a.tab[q & CMASK], a.tab[(q >> BPS) & CMASK],
                             a.tab[(q >> (2 * BPS)) & CMASK], a.tab[(q >> (3 * BPS)) & CMASK]};
        }
    }
    __syncthreads();
    const int lane = bb_lane();
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    const uint64_t E = a.ndw * (32 / BPS);
    const uint64_t nwork = a.nfs * a.nseg;
    const uint64_t pbytes = a.ndw * 4;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const uint8_t *stage8 = reinterpret_cast<const uint8_t *>(&s_stage[wave][0]);
    uint32_t *stage32 = reinterpret_cast<uint32_t *>(&s_stage[wave][0]);
    // byte of the tile a lane's float4 comes from in pass p: 1-bit 32 p + l / 2,
    // 2-bit 64 p + l, 4-bit two bytes at 128 p + 2 l
    const uint32_t lane_byte = BPS == 1 ? (uint32_t)lane >> 1 : BPS == 2 ? (uint32_t)lane
                             : BPS == 4 ? 2u * (uint32_t)lane : 4u * (uint32_t)lane;
    const uint32_t hsel = (uint32_t)(lane % FPB);

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = work; seg = 0; }
        else { fs = work / a.nseg; seg = work - fs * a.nseg; }
        const int64_t so = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t seg_b_end = (seg + 1) * a.seg_tiles * 256 < pbytes ? (seg + 1) * a.seg_tiles * 256 : pbytes;
        const uint64_t b0 = tile0 * 256;                                  // first payload byte of this wave
        uint64_t nb = 0;                                                  // bytes this wave decodes
        if (b0 < seg_b_end) { nb = seg_b_end - b0; if (nb > (uint64_t)a.tpw * 256) nb = (uint64_t)a.tpw * 256; }
        float *obase = a.out + bb_out_slot(a, fs) * E + b0 * (8 / BPS);
        uint32_t s = 0;
        if (valid && nb && (reinterpret_cast<uintptr_t>(a.buf + (uint64_t)so) & 3)) {
            // a payload at an odd address: byte loads, staged from offset 0
            const uint8_t *pp = a.buf + (uint64_t)so + b0;
            uint8_t *st8 = reinterpret_cast<uint8_t *>(&s_stage[wave][0]);
#pragma nounroll
            for (uint32_t i = (uint32_t)lane; i < (uint32_t)nb; i += BB_WAVE) st8[i] = pp[i];
        } else if (valid && nb) {
            const uint8_t *pp = a.buf + (uint64_t)so + b0;                // 4-byte aligned
            s = (uint32_t)(reinterpret_cast<uintptr_t>(pp) & 255);
            const uint8_t *base = pp - s;                                 // 256-byte aligned address
            const uint32_t lo = s, hi = s + (uint32_t)nb;                 // wanted bytes of the staged image
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) {
                const uint32_t piece = (uint32_t)k * BB_WAVE + (uint32_t)lane;
                const uint32_t p0 = piece * 16;
                if (piece >= (uint32_t)NPIECE || p0 + 16 <= lo || p0 >= hi) continue;
                if (p0 >= lo && p0 + 16 <= hi) {
                    if (GL)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(base + p0),
                            (__attribute__((address_space(3))) void *)(&s_stage[wave][k * BB_WAVE]), 16, 0, AUX);
                    else
                        s_stage[wave][piece] = *reinterpret_cast<const bb_u4 *>(base + p0);
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t q = p0 + 4 * d;
                        if (q >= lo && q + 4 <= hi) stage32[piece * 4 + d] = *reinterpret_cast<const uint32_t *>(base + q);
                    }
                }
            }
        }
        if (GL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the direct-to-LDS loads have landed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t e_end = nb * (8 / BPS);                            // elements of this wave
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            if ((uint64_t)u * EPT >= e_end) break;
            const bool ragged = (uint64_t)(u + 1) * EPT > e_end;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const uint32_t e = (uint32_t)u * EPT + 256 * p + 4 * lane;
                if (ragged && e >= e_end) continue;
                bb_f4 v;
                if (!valid) {
                    v = fillv;
                } else {
                    const uint32_t b = s + (uint32_t)u * 256 + (uint32_t)(256 / PASSES) * p + lane_byte;
                    if (BPS == 8) {
                        const uint32_t w = stage32[b >> 2];         // (s and lane_byte are multiples of 4)
                        if (LV == BB_LV_LDS)
                            v = bb_f4{s_tab8[w & 0xff], s_tab8[(w >> 8) & 0xff], s_tab8[(w >> 16) & 0xff], s_tab8[w >> 24]};
                        else
                            v = bb_f4{(float)(int)(int8_t)(w & 0xff), (float)(int)(int8_t)((w >> 8) & 0xff),
                                      (float)(int)(int8_t)((w >> 16) & 0xff), (float)(int)(int8_t)(w >> 24)};
                    } else if (BPS == 4) {
                        const float2 l2 = s_lut2[stage8[b]], h2 = s_lut2[stage8[b + 1]];
                        v = bb_f4{l2.x, l2.y, h2.x, h2.y};
                    } else {
                        v = s_lut[(uint32_t)stage8[b] * FPB + hsel];
                    }
                }
                bb_store4<NT>(obase + e, v);
            }
        }
        __builtin_amdgcn_wave_barrier();          // the next step overwrites the staging area
    }
}
