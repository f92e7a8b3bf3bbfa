// Folded channel subsets of thread-interleaved frames, one work item per WAVE
// (round 5).
//
// Replaces (reference, path:line) the `subset` indexing of the decoded frame
// set (base/base.py:706-717, vdif/base.py:519-528) on top of the thread gather
// of VDIFFrameSet.__getitem__ (vdif/frame.py:402-434): only the kept positions
// of every thread sample are decoded and written, in (sample, thread, kept)
// order -- k_decode_gather_select's job, restructured the way k_decode_flat_lds
// restructured the flat decode in round 4:
//
//   * a work item belongs to ONE wave: SB consecutive payload bytes of every
//     thread slot of a frame set (SB = 1 KiB for 8 slots), staged in the
//     wave's own piece of LDS -- no workgroup barrier anywhere;
//   * the bytes go HBM -> LDS directly, global_load_lds_dwordx4: one wave
//     instruction moves 1 KiB of a slot's payload, no VGPR round trip (ragged
//     ends and payloads off 16-byte alignment by dword, odd addresses by byte);
//   * the output of an item is contiguous (SB / rowbytes time samples x nslot
//     x nsel floats); a lane's float4 sits at the SAME place of its output row
//     in every pass (rowlen divides 256), so which slot and which bits each of
//     its four floats comes from is worked out once per kernel; per float4:
//     one LDS byte per distinct byte (the re / im of a complex channel share
//     theirs), the level, one 16-byte nt store.
//
// Conditions (the host falls back to k_decode_gather_select otherwise): a
// thread sample is whole bytes (bps * chunk % 8 == 0); nslot * nsel is a power
// of two of 4 .. 256; nslot <= 32.
#pragma once
#include "bb_common.h"
#include "k_flat.h"

struct bb_pick_args {
    const uint8_t *buf;
    const int64_t *src;     // [nframes * nslot], -1 = fill
    float         *out;
    const float   *tab;
    const int32_t *within;  // [nsel] kept positions of a thread sample
    uint64_t nframes;       // frame sets
    uint64_t pbytes;        // payload bytes per slot
    uint64_t src_lim;       // offsets outside [0, src_lim) decode as fill (bb_src_ok)
    uint32_t nslot, nsel;
    uint32_t lrowlen;       // log2(nslot * nsel): floats per output row
    uint32_t rowbytes;      // bytes of one thread sample
    uint32_t sb;            // payload bytes per slot staged per item (multiple of rowbytes and of 16)
    uint32_t nitem;         // items per frame set
    uint32_t pitch;         // bytes between the slots' rows of a wave's stage
    float    fill_re, fill_im;
    int32_t  complex_data;
    bb_perm_t perm;
};

template <int BPS, int LV, bool NT, int NW>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_pick(bb_pick_args a)
{
    constexpr int NCODE = 1 << BPS;
    constexpr uint32_t CMASK = NCODE - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_pick[];
    __shared__ float s_tab[LV == BB_LV_LDS ? NCODE : 1];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (LV == BB_LV_LDS) {
        for (int i = threadIdx.x; i < NCODE; i += NW * BB_WAVE) s_tab[i] = a.tab[i];
        __syncthreads();
    } else if (LV == BB_LV_REG) {
        t0 = a.tab[0]; t1 = a.tab[1];
        if (BPS == 2) { t2 = a.tab[2]; t3 = a.tab[3]; }
    }
    auto level = [&](uint32_t code) -> float {
        if constexpr (LV == BB_LV_REG) {
            if (BPS == 1) return code ? t1 : t0;
            const float lo = (code & 1) ? t1 : t0;
            const float hi = (code & 1) ? t3 : t2;
            return (code & 2) ? hi : lo;
        } else if constexpr (LV == BB_LV_LDS) {
            return s_tab[code];
        } else {
            return (float)(int)(int8_t)code;
        }
    };
    const int lane = bb_lane();
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    uint8_t *stage = s_pick + (size_t)wave * a.nslot * a.pitch;
    const uint32_t rowlen = 1u << a.lrowlen;

    // the lane's four floats: slot, byte within the thread sample, shift, fill
    const uint32_t rem = ((uint32_t)lane * 4) & (rowlen - 1);
    uint32_t slot[4], bo[4], sh[4];
    float fl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t pos = rem + j;
        slot[j] = pos / a.nsel;
        const uint32_t within = (uint32_t)a.within[pos - slot[j] * a.nsel];
        bo[j] = slot[j] * a.pitch + ((within * BPS) >> 3);
        sh[j] = (within * BPS) & 7;
        fl[j] = (a.complex_data && (within & 1)) ? a.fill_im : a.fill_re;
    }
    const bool same1 = bo[1] == bo[0], same2 = bo[2] == bo[1], same3 = bo[3] == bo[2];
    const uint32_t row_step = 256u >> a.lrowlen;                        // rows a wave pass advances by
    const uint32_t row0 = ((uint32_t)lane * 4) >> a.lrowlen;            // the lane's row in pass 0
    const uint64_t R = a.pbytes / a.rowbytes;                           // rows per frame set
    const uint64_t nwork = a.nframes * a.nitem;

    for (uint64_t w = (uint64_t)blockIdx.x * NW + wave; w < nwork; w += (uint64_t)gridDim.x * NW) {
        const uint64_t work = bb_perm(a.perm, w);
        const uint64_t f = work / a.nitem;
        const uint32_t it = (uint32_t)(work - f * a.nitem);
        const uint64_t byte0 = (uint64_t)it * a.sb;
        const uint32_t nb = (uint32_t)((a.pbytes - byte0 < a.sb) ? a.pbytes - byte0 : a.sb);
        // lane s holds slot s's payload offset (nslot <= 32)
        const int64_t my_so = (uint32_t)lane < a.nslot ? a.src[f * a.nslot + (uint32_t)lane] : -1;
        const bool my_ok = bb_src_ok(my_so, a.src_lim);
        const uint64_t my_ad = reinterpret_cast<uintptr_t>(a.buf + (my_ok ? (uint64_t)my_so : 0) + byte0);
        // where in its row the slot's first wanted byte lands: the address's offset
        // in its 16-byte piece (0 for odd addresses, which are copied by bytes)
        const uint32_t my_mis = (my_ad & 3) ? 0u : (uint32_t)(my_ad & 15);
        for (uint32_t s = 0; s < a.nslot; ++s) {
            const int ok = __shfl((int)my_ok, (int)s);
            if (!ok) continue;                                          // wave-uniform
            const uint32_t lo32 = (uint32_t)__shfl((int)(uint32_t)(my_ad & 0xffffffffu), (int)s);
            const uint32_t hi32 = (uint32_t)__shfl((int)(uint32_t)(my_ad >> 32), (int)s);
            const uint8_t *pp = reinterpret_cast<const uint8_t *>(((uint64_t)hi32 << 32) | lo32);
            uint8_t *row = stage + s * a.pitch;
            if (lo32 & 3) {
                for (uint32_t i = (uint32_t)lane; i < nb; i += BB_WAVE) row[i] = pp[i];
                continue;
            }
            const uint32_t mis = lo32 & 15;
            const uint8_t *base = pp - mis;                             // 16-byte aligned
            const uint32_t lo = mis, hi = mis + nb;                     // wanted bytes of the staged image
            for (uint32_t k0 = 0; k0 * 16 < hi; k0 += BB_WAVE) {        // wave-uniform trip count
                const uint32_t p0 = (k0 + (uint32_t)lane) * 16;
                if (p0 + 16 <= lo || p0 >= hi) continue;
                if (p0 >= lo && p0 + 16 <= hi) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + p0),
                                                     (__attribute__((address_space(3))) void *)(row + k0 * 16), 16, 0, 0);
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t q = p0 + 4 * d;
                        if (q >= lo && q + 4 <= hi)
                            *reinterpret_cast<uint32_t *>(row + q) = *reinterpret_cast<const uint32_t *>(base + q);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the direct-to-LDS loads have landed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t off[4];
        bool okj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            off[j] = bo[j] + (uint32_t)__shfl((int)my_mis, (int)slot[j]);
            okj[j] = __shfl((int)my_ok, (int)slot[j]) != 0;
        }
        const uint32_t nrow = nb / a.rowbytes;
        const uint32_t nfloat = nrow << a.lrowlen;
        float *obase = a.out + ((f * R + byte0 / a.rowbytes) << a.lrowlen);
        uint32_t rb = row0 * a.rowbytes;
        const uint32_t rb_step = row_step * a.rowbytes;
        for (uint32_t q = (uint32_t)lane * 4; q < nfloat; q += 256, rb += rb_step) {
            uint32_t b[4];
            b[0] = stage[rb + off[0]];
            b[1] = same1 ? b[0] : (uint32_t)stage[rb + off[1]];
            b[2] = same2 ? b[1] : (uint32_t)stage[rb + off[2]];
            b[3] = same3 ? b[2] : (uint32_t)stage[rb + off[3]];
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t code = BPS == 8 ? b[j] : ((b[j] >> sh[j]) & CMASK);
                r[j] = okj[j] ? level(code) : fl[j];
            }
            bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
        }
        __builtin_amdgcn_wave_barrier();                                // the next item overwrites the stage
    }
}
