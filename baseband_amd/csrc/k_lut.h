// Flat decode of 1-, 2- and 4-bit samples through a BYTE table in LDS.
//
// Replaces the same reference expressions as k_decode_flat_aln -- and in the
// same way the reference does it: lut.take(bytes) with a 256-row table of the
// 8 / bps floats a byte holds (vdif/payload.py:25-103, mark5b/payload.py:27-94).
//
// Why: the ISA of k_decode_flat_aln<2,...> has 97 instructions per 1 KiB store
// (13 v_cndmask + 8 v_cmp + 7 v_and for the register level select of four
// samples, 8 s_nop, 5 branches and 10 exec-mask operations for the per-pass
// range checks, 64-bit address arithmetic), and its waves spend half their
// cycles stalled on instruction issue (profiles/r02q_8GiB_kernels.csv).  Here a
// store costs: one ds_bpermute (two when the payload is not 256-byte aligned),
// a bit-field extract of the lane's byte, ONE ds_read_b128 of that byte's four
// floats (2-bit; for 1-bit data a byte holds two float4), the address add and
// the store; range checks are per TILE and wave-uniform (only the last, ragged
// tile of a payload checks per pass), and the aligned / misaligned hand-out are
// two loop bodies chosen per work item.
//
// Geometry, persistence, one-item-ahead loads, aligned blocks and the striped
// work order are those of k_decode_flat_aln.
#pragma once
#include "k_flat.h"

template <int BPS, bool NT, int NW, int TPW>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_flat_lut(bb_flat_args a)
{
    static_assert(BPS == 1 || BPS == 2 || BPS == 4, "byte table kernel: 1-, 2- or 4-bit samples");
    constexpr int EPT = 2048 / BPS;             // elements per 256-byte tile
    constexpr int PASSES = 8 / BPS;             // 1 KiB store passes per tile
    constexpr int FPB = BPS == 1 ? 2 : 1;       // float4 per input byte: 1 (2-bit), 2 (1-bit); 4-bit: below
    constexpr uint32_t CMASK = (1u << BPS) - 1;
    // 1-/2-bit: s_lut[byte * FPB + h] = the h-th float4 of that byte.
    // 4-bit: a byte holds two samples -- s_lut2[byte] = (low nibble, high nibble),
    // and a lane's float4 is two bytes, two 8-byte table reads.
    __shared__ bb_f4 s_lut[BPS == 4 ? 1 : 256 * FPB];
    __shared__ float2 s_lut2[BPS == 4 ? 256 : 1];
    if (BPS == 4) {
        for (int i = threadIdx.x; i < 256; i += NW * BB_WAVE)
            s_lut2[i] = float2{a.tab[i & 15], a.tab[i >> 4]};
    } else {
        for (int i = threadIdx.x; i < 256 * FPB; i += NW * BB_WAVE) {
            const uint32_t b = (uint32_t)i / FPB, h = (uint32_t)i % FPB;
            const uint32_t q = b >> (4 * BPS * h);          // the four codes of this float4
            s_lut[i] = bb_f4{a.tab[q & CMASK], a.tab[(q >> BPS) & CMASK],
                             a.tab[(q >> (2 * BPS)) & CMASK], a.tab[(q >> (3 * BPS)) & CMASK]};
        }
    }
    __syncthreads();
    const int lane = bb_lane();
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    const uint64_t E = a.ndw * (32 / BPS);
    const uint64_t nwork = a.nfs * a.nseg;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    // pass p: lane l writes float4 number 64 p + l of the tile; its 4 BPS code
    // bits start at bit 4 BPS (64 p + l) of the tile: dword 8 BPS p + l BPS / 8,
    // byte (1-/2-bit) or halfword (4-bit) of that dword below
    const int src_lane0 = (lane * BPS) >> 3;
    const uint32_t bshift = BPS == 4 ? (uint32_t)(lane & 1) * 16
                                     : (uint32_t)((lane / FPB) & 3) * 8;
    const uint32_t hsel = (uint32_t)(lane % FPB);                   // which float4 of the byte (1-bit)

    uint32_t cur[TPW + 1], nxt[TPW + 1];
    bool cur_valid = false, nxt_valid = false;
    uint32_t cur_s = 0, nxt_s = 0;

    auto issue = [&](uint64_t step, uint32_t (&w)[TPW + 1], bool &valid, uint32_t &s) {
        const uint64_t work = bb_perm(a.perm, step);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = work; seg = 0; }
        else { fs = work / a.nseg; seg = work - fs * a.nseg; }
        const int64_t so = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
        valid = bb_src_ok(so, a.src_lim);
        const uint8_t *pp = a.buf + (valid ? (uint64_t)so : 0);
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(pp);
        s = (b0 & 3) ? 0u : (uint32_t)((b0 >> 2) & 63);    // odd byte addresses: plain loads
        const uint32_t *blk = reinterpret_cast<const uint32_t *>(pp) - s;
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t dw_end = (seg + 1) * a.seg_tiles * 64 < a.ndw
                                ? (seg + 1) * a.seg_tiles * 64 : a.ndw;
#pragma unroll
        for (int u = 0; u <= TPW; ++u) {
            const uint64_t j = (tile0 + u) * 64 + lane;     // block dword j holds payload dword j - s
            const bool want = valid && u <= (int)a.tpw && j >= s && j - s < dw_end;
            w[u] = want ? bb_load_dw(a, &blk[j]) : 0u;
        }
    };

    // one tile: PASSES stores of 1 KiB; ALIGNED: the payload starts on a
    // 256-byte block (one shuffle); CHECK: the tile may end inside a pass
    auto tile_out = [&](auto aligned_tag, auto check_tag, uint32_t w0, uint32_t w1, uint32_t s,
                        float *otile, uint64_t e_tile, uint64_t e_end) {
        constexpr bool ALIGNED = decltype(aligned_tag)::value;
        constexpr bool CHECK = decltype(check_tag)::value;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t idx = (uint32_t)(p * 8 * BPS + src_lane0) + (ALIGNED ? 0u : s);
            uint32_t word = (uint32_t)__shfl((int)w0, (int)(idx & 63));
            if (!ALIGNED) {
                const uint32_t hi = (uint32_t)__shfl((int)w1, (int)(idx & 63));
                word = idx >= 64 ? hi : word;
            }
            bb_f4 v;
            if (BPS == 4) {
                const uint32_t hw = word >> bshift;
                const float2 lo = s_lut2[hw & 0xffu], hi = s_lut2[(hw >> 8) & 0xffu];
                v = bb_f4{lo.x, lo.y, hi.x, hi.y};
            } else {
                v = s_lut[((word >> bshift) & 0xffu) * FPB + hsel];
            }
            if (CHECK && e_tile + 256 * p + 4 * lane >= e_end) continue;
            bb_store4<NT>(otile + 256 * p + 4 * lane, v);
        }
    };

    uint64_t step = blockIdx.x;
    if (step < nwork) issue(step, cur, cur_valid, cur_s);
    for (; step < nwork; step += gridDim.x) {
        const uint64_t next = step + gridDim.x;
        if (next < nwork) issue(next, nxt, nxt_valid, nxt_s);

        const uint64_t pwork = bb_perm(a.perm, step);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = pwork; seg = 0; }
        else { fs = pwork / a.nseg; seg = pwork - fs * a.nseg; }
        float *obase = a.out + bb_out_slot(a, fs) * E;
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t seg_e_end = (seg + 1) * a.seg_tiles * EPT < E
                                   ? (seg + 1) * a.seg_tiles * EPT : E;
        // tiles of this wave that lie completely / partly below seg_e_end
        const uint64_t e0w = tile0 * EPT;
        uint32_t nfull = 0, npart = 0;
        if (e0w < seg_e_end) {
            const uint64_t left = seg_e_end - e0w;
            nfull = (uint32_t)(left / EPT);
            if (nfull > a.tpw) nfull = a.tpw;
            npart = (nfull < a.tpw && left > (uint64_t)nfull * EPT) ? 1u : 0u;
        }
        float *owave = obase + e0w;
        if (!cur_valid) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if (u >= (int)(nfull + npart)) break;
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const uint64_t e = e0w + (uint64_t)u * EPT + 256 * p + 4 * lane;
                    if (e < seg_e_end) bb_store4<NT>(obase + e, fillv);
                }
            }
        } else if (cur_s == 0) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if (u < (int)nfull)
                    tile_out(std::true_type{}, std::false_type{}, cur[u], 0u, 0u, owave + u * EPT, 0, 0);
                else if (u == (int)nfull && npart)
                    tile_out(std::true_type{}, std::true_type{}, cur[u], 0u, 0u, owave + u * EPT,
                             e0w + (uint64_t)u * EPT, seg_e_end);
            }
        } else {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if (u < (int)nfull)
                    tile_out(std::false_type{}, std::false_type{}, cur[u], cur[u + 1], cur_s, owave + u * EPT, 0, 0);
                else if (u == (int)nfull && npart)
                    tile_out(std::false_type{}, std::true_type{}, cur[u], cur[u + 1], cur_s, owave + u * EPT,
                             e0w + (uint64_t)u * EPT, seg_e_end);
            }
        }
#pragma unroll
        for (int u = 0; u <= TPW; ++u) cur[u] = nxt[u];
        cur_valid = nxt_valid;
        cur_s = nxt_s;
    }
}
