// Thread interleave for narrow samples: frame sets whose per-thread chunk is
// smaller than one float4 (nchan * ncomp = 1 or 2, e.g. the 8-thread,
// 1-channel layout of sample.vdif).
//
// Replaces (reference, path:line) the same expressions as k_decode_flat plus
// the strided per-thread copies of VDIFFrameSet.__getitem__
// (vdif/frame.py:427-434): there every thread's (nsample, nchan) block is
// written into column t of a (nsample, nthread, nchan) array; here the output
// rows are assembled on chip so that HBM only sees contiguous 16-byte stores.
//
// A workgroup stages G tiles (256 bytes each) of the SAME position from every
// thread slot's payload in LDS (coalesced dword loads), then walks the
// contiguous output region those tiles map to: lane -> float4 -> four
// (row, slot) pairs -> one LDS byte read + bit-field extract each.
#pragma once
#include "bb_common.h"
#include "k_flat.h"

struct bb_gather_args {
    const uint8_t *buf;
    const int64_t *src;     // [nframes * nslot], -1 = fill; never null here
    float         *out;
    const float   *tab;
    uint64_t nframes;
    uint64_t ndw;           // payload dwords per slot
    uint32_t nslot, chunk, lchunk;
    uint32_t gtiles;        // tiles per slot staged per work item
    uint32_t ngroup;        // work items per frame set
    float    fill_re, fill_im;
    int32_t  complex_data;
    int32_t  lrow;          // log2(nslot * chunk) when that is a power of two, else -1
    int32_t  aligned;       // use 256-byte aligned block loads
    int32_t  glds;          // stage with direct-to-LDS loads (global_load_lds_dword) instead of load + ds_write
    uint64_t src_lim;       // offsets outside [0, src_lim) decode as fill (bb_src_ok)
    bb_perm_t perm;         // work order (bb_common.h)
    // channel selection (bb_decode_frames_select): only positions within[0..nsel)
    // of every thread sample's chunk are written, in that order; nsel == 0: all
    const int32_t *within;
    uint32_t nsel;
    // floor(2^32 / d) + 1 for d = nslot * nsel and d = nsel: q / d ==
    // __umulhi(q, magic) for every q with q * d < 2^32 (the host checks the
    // bound); 0 = divide (d == 1, or the bound does not hold)
    uint32_t mag_row, mag_sel;
};

__device__ __forceinline__ uint32_t bb_div_magic(uint32_t q, uint32_t d, uint32_t magic)
{
    return magic ? __umulhi(q, magic) : q / d;
}

// WIDE: chunks of at least four floats (a float4 never straddles thread slots)
// Phase 1 of the gather kernels: the group's dwords of every thread slot ->
// LDS rows (256-byte aligned block loads; the misalignment of a payload against
// the blocks, in dwords, goes into s_base).  All payload offsets of the frame
// set are fetched with ONE coalesced load per 64 slots and handed out with
// shuffles, and the payload loads of different slots are issued eight at a
// time before their LDS writes: with one load chain per slot (offset, then
// payload, then the next slot's offset ...) 64 slots took 16 dependent round
// trips per wave and the kernel ran at 3.6 TB/s (profiles/r02o_exp_gather64_before.log).
__device__ __forceinline__ void bb_gather_stage(const bb_gather_args &a, uint64_t f, uint64_t dw0,
                                                uint32_t gdw, uint32_t pitch, uint32_t *s_raw,
                                                uint32_t *s_valid, uint32_t *s_base, uint32_t *s_missing)
{
    const int lane = bb_lane();
    const uint32_t wave = (uint32_t)bb_wave();
    // one wave per slot at a time; with one or two slots the waves share a slot's pieces
    const uint32_t wps = a.nslot == 1 ? 4u : (a.nslot == 2 ? 2u : 1u);
    const uint32_t wsub = wave % wps, sfirst = wave / wps, sstep = BB_WAVES_PER_BLOCK / wps;
    const uint32_t nblk = (gdw + 64 + 64 * wps - 1) / (64 * wps);      // 64-dword pieces per slot and wave
    for (uint32_t s0 = 0; s0 < a.nslot; s0 += BB_WAVE) {
        const uint32_t sl = s0 + (uint32_t)lane;
        const int64_t my_so = sl < a.nslot ? a.src[f * a.nslot + sl] : -1;
        const bool my_ok = bb_src_ok(my_so, a.src_lim);
        const uint8_t *my_p = a.buf + (my_ok ? (uint64_t)my_so : 0);
        const uintptr_t my_ad = reinterpret_cast<uintptr_t>(my_p);
        // misalignment of the payload against 256-byte blocks of the address
        // space, in dwords (odd byte addresses keep plain loads)
        const uint32_t my_sh = (a.aligned && !(my_ad & 3)) ? (uint32_t)((my_ad >> 2) & 63) : 0u;
        if (sl < a.nslot && wave == 0) {
            s_valid[sl] = my_ok ? 1u : 0u;
            s_base[sl] = (sl * pitch + my_sh) * 4;
            if (s_missing && !my_ok) atomicAdd(s_missing, 1u);
        }
        const uint32_t send = (s0 + BB_WAVE < a.nslot) ? s0 + BB_WAVE : a.nslot;
        // this wave's slots in [s0, send): sfirst + k * sstep
        const uint32_t k0 = s0 > sfirst ? (s0 - sfirst + sstep - 1) / sstep : 0u;
        const uint32_t sbeg = sfirst + k0 * sstep;
        const uint32_t ns = sbeg < send ? (send - sbeg + sstep - 1) / sstep : 0u;
        const uint32_t npiece = ns * nblk;
        if (a.glds) {
            // Direct-to-LDS (round 4): a piece = 64 consecutive dwords of one slot = one
            // global_load_lds_dword per wave, landing lane-linear in the slot's row; no
            // VGPR round trip, no ds_write, all pieces in flight at once (the barrier
            // behind the staging waits for them).  Dwords outside the payload are not
            // loaded (and never decoded).  Payloads at odd addresses keep the path below.
            for (uint32_t p = 0; p < npiece; ++p) {
                const uint32_t k = p / nblk, b = p - k * nblk;
                const uint32_t s = sbeg + k * sstep;
                const int from = (int)(s - s0);
                const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)((uint64_t)my_ad & 0xffffffffu), from);
                const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)((uint64_t)my_ad >> 32), from);
                const uint32_t sh = (uint32_t)__shfl((int)my_sh, from);
                const int ok = __shfl((int)my_ok, from);
                const uint32_t *blk = reinterpret_cast<const uint32_t *>(((uint64_t)hi << 32) | lo) - sh;
                const uint32_t jb = (b * wps + wsub) * BB_WAVE;           // wave-uniform
                const uint32_t j = jb + (uint32_t)lane;
                const uint64_t q = dw0 + j;
                if (!ok || jb >= gdw + 64) continue;                       // wave-uniform
                if (lo & 3) {                                              // odd address: through registers
                    if (j < gdw + 64 && q >= sh && q - sh < a.ndw) s_raw[s * pitch + j] = blk[q];
                    continue;
                }
                if (j < gdw + 64 && q >= sh && q - sh < a.ndw)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + q),
                                                     (__attribute__((address_space(3))) void *)(s_raw + s * pitch + jb), 4, 0, 0);
            }
            // LDS-DMA loads are counted by vmcnt, which a barrier does not wait for
            // by itself: other waves read what this wave staged (as k_lds.h, k_tfpick.h)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        for (uint32_t p0 = 0; p0 < npiece; p0 += 8) {
            uint32_t r[8], dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t p = p0 + u;
                r[u] = 0u; dst[u] = 0xffffffffu;
                if (p < npiece) {                               // wave-uniform
                    const uint32_t k = p / nblk, b = p - k * nblk;
                    const uint32_t s = sbeg + k * sstep;
                    const int from = (int)(s - s0);
                    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)((uint64_t)my_ad & 0xffffffffu), from);
                    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)((uint64_t)my_ad >> 32), from);
                    const uint32_t sh = (uint32_t)__shfl((int)my_sh, from);
                    const int ok = __shfl((int)my_ok, from);
                    const uint32_t *blk = reinterpret_cast<const uint32_t *>(((uint64_t)hi << 32) | lo) - sh;
                    const uint32_t j = (b * wps + wsub) * BB_WAVE + (uint32_t)lane;
                    const uint64_t q = dw0 + j;                 // block dword q = payload dword q - sh
                    if (j < gdw + 64) {
                        dst[u] = s * pitch + j;
                        if (ok && q >= sh && q - sh < a.ndw) r[u] = blk[q];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (dst[u] != 0xffffffffu) s_raw[dst[u]] = r[u];
        }
    }
}

// The same staging with a CHANNEL SELECTION folded in (reader `subset`: the
// reference decodes whole frames and indexes the result afterwards,
// base/base.py:706-717, 957-969): only positions within[0 .. nsel) of every
// thread sample are written, so the output -- and its HBM traffic -- shrinks to
// nsel / chunk of the full decode and no second pass over it is needed.  V4:
// every work item's share of the output starts on a 16-byte boundary and is a
// multiple of four floats (the host checks), so lanes write float4 whatever
// the selection is; otherwise one float per lane and step.
template <int BPS, int LV, bool NT, bool V4>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_gather_select(bb_gather_args a)
{
    constexpr int NCODE = 1 << BPS;
    constexpr uint32_t CMASK = NCODE - 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    const uint32_t pitch = a.gtiles * 64 + 64 + 1;
    uint32_t *s_raw = s_mem;
    uint32_t *s_valid = s_mem + (size_t)a.nslot * pitch;
    uint32_t *s_base = s_valid + a.nslot;
    float *s_tab = reinterpret_cast<float *>(s_base + a.nslot + 1);
    uint32_t *s_within = reinterpret_cast<uint32_t *>(s_tab + (LV == BB_LV_LDS ? NCODE : 0));
    __shared__ float s_fill[2];
    if (threadIdx.x == 0) { s_fill[0] = a.fill_re; s_fill[1] = a.fill_im; }

    // (levels as plain locals, not the bb_levels struct: captured by reference
    // by the lambda below, the struct was kept on the stack in the table
    // variants -- 16 bytes of scratch per lane)
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (LV == BB_LV_LDS) {
        for (int i = threadIdx.x; i < NCODE; i += BB_BLOCK) s_tab[i] = a.tab[i];
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
    for (uint32_t i = threadIdx.x; i < a.nsel; i += BB_BLOCK) s_within[i] = (uint32_t)a.within[i];
    const uint64_t E = a.ndw * (32 / BPS);
    const uint64_t R = E >> a.lchunk;
    const uint32_t rowlen = a.nslot * a.nsel;               // floats per output row
    const uint64_t nwork = a.nframes * a.ngroup;
    const uint32_t gdw = a.gtiles * 64;

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / a.ngroup;
        const uint32_t g = (uint32_t)(work - f * a.ngroup);
        const uint64_t dw0 = (uint64_t)g * gdw;
        __syncthreads();
        bb_gather_stage(a, f, dw0, gdw, pitch, s_raw, s_valid, s_base, nullptr);
        __syncthreads();
        const uint64_t e_lo = dw0 * (32 / BPS);
        const uint64_t e_hi = (e_lo + (uint64_t)gdw * (32 / BPS) < E) ? e_lo + (uint64_t)gdw * (32 / BPS) : E;
        const uint32_t nrow = (uint32_t)((e_hi - e_lo) >> a.lchunk);
        const uint32_t nfloat = nrow * rowlen;
        float *obase = a.out + (f * R + (e_lo >> a.lchunk)) * rowlen;
        const uint8_t *rawb = reinterpret_cast<const uint8_t *>(s_raw);
        auto value = [&](uint32_t row, uint32_t s, uint32_t k) -> float {
            const uint32_t within = s_within[k];
            const uint32_t bit = ((row << a.lchunk) + within) * BPS;
            uint32_t code;
            if (BPS == 8) code = rawb[s_base[s] + (bit >> 3)];
            else code = ((uint32_t)rawb[s_base[s] + (bit >> 3)] >> (bit & 7)) & CMASK;
            float v = level(code);
            // (the fill values wait in LDS: held in SGPRs across the kernel they were
            // the two values the table variants spilled to scratch)
            if (!s_valid[s]) v = s_fill[(a.complex_data && (within & 1)) ? 1 : 0];
            return v;
        };
        if (V4) {
            // four consecutive floats of the (contiguous) output per lane; they
            // may belong to different thread slots and rows: the position walks
            // on with carries (k -> slot -> row), one index division per float4
            if ((a.nsel & 3) == 0 && (rowlen & (rowlen - 1)) == 0 && rowlen <= BB_BLOCK * 4) {
                // A power-of-two output row of at most 1024 floats (8 threads x 2 of 16
                // complex channels: 32): the workgroup advances by 1024 floats per pass,
                // so a lane's float4 sits at the SAME place of its row in every pass --
                // slot, kept positions, that slot's LDS base and validity are looked up
                // once per work item; per float only the row's bit offset, one LDS byte
                // and the level remain (round 4: 0.59 -> 0.7x of the peak on bytes moved)
                const uint32_t rem = (threadIdx.x * 4) & (rowlen - 1);
                const uint32_t s = bb_div_magic(rem, a.nsel, a.mag_sel), k = rem - s * a.nsel;
                const uint32_t base = s_base[s];
                const bool ok = s_valid[s] != 0;
                uint32_t wb[4];
                float fl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t within = s_within[k + j];
                    wb[j] = within * BPS;
                    fl[j] = s_fill[(a.complex_data && (within & 1)) ? 1 : 0];
                }
                const uint32_t lrow = 31u - (uint32_t)__clz((int)rowlen);
                if (BPS != 8 && ((BPS << a.lchunk) & 7) == 0) {
                    // rows of whole bytes (every multi-channel layout): a kept value's
                    // byte within its row and its shift are fixed per lane, and values
                    // that share a byte -- the re / im of a complex channel always do --
                    // share one LDS read: 2 instead of 4 ds_read_u8 per float4 for two
                    // complex channels, 1 for a neighbouring aligned pair
                    const uint32_t rowbytes = (uint32_t)(BPS << a.lchunk) >> 3;
                    uint32_t bo[4], sh[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bo[j] = base + (wb[j] >> 3); sh[j] = wb[j] & 7; }
                    const bool s1 = bo[1] == bo[0], s2 = bo[2] == bo[1], s3 = bo[3] == bo[2];
                    for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
                        const uint32_t rb = (q >> lrow) * rowbytes;
                        uint32_t b[4];
                        b[0] = rawb[rb + bo[0]];
                        b[1] = s1 ? b[0] : (uint32_t)rawb[rb + bo[1]];
                        b[2] = s2 ? b[1] : (uint32_t)rawb[rb + bo[2]];
                        b[3] = s3 ? b[2] : (uint32_t)rawb[rb + bo[3]];
                        float r[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = ok ? level((b[j] >> sh[j]) & CMASK) : fl[j];
                        bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
                    }
                } else
                for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
                    const uint32_t rbit = ((q >> lrow) << a.lchunk) * BPS;
                    float r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t bit = rbit + wb[j];
                        uint32_t code = rawb[base + (bit >> 3)];
                        if (BPS != 8) code = (code >> (bit & 7)) & CMASK;
                        r[j] = ok ? level(code) : fl[j];
                    }
                    bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
                }
            } else if ((a.nsel & 3) == 0) {
                // (a selection of whole float4s: the four floats share their slot and row)
                for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
                    const uint32_t row = bb_div_magic(q, rowlen, a.mag_row), rem = q - row * rowlen;
                    const uint32_t s = bb_div_magic(rem, a.nsel, a.mag_sel), k = rem - s * a.nsel;
                    bb_store4<NT>(obase + q, bb_f4{value(row, s, k), value(row, s, k + 1),
                                                   value(row, s, k + 2), value(row, s, k + 3)});
                }
            } else
            for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
                uint32_t row = bb_div_magic(q, rowlen, a.mag_row);
                const uint32_t rem = q - row * rowlen;
                uint32_t s = bb_div_magic(rem, a.nsel, a.mag_sel), k = rem - s * a.nsel;
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r[j] = value(row, s, k);
                    if (++k == a.nsel) { k = 0; if (++s == a.nslot) { s = 0; ++row; } }
                }
                bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
            }
        } else {
            for (uint32_t q = threadIdx.x; q < nfloat; q += BB_BLOCK) {
                const uint32_t row = bb_div_magic(q, rowlen, a.mag_row), rem = q - row * rowlen;
                const uint32_t s = bb_div_magic(rem, a.nsel, a.mag_sel);
                bb_store1<NT>(obase + q, value(row, s, rem - s * a.nsel));
            }
        }
    }
}

template <int BPS, int LV, bool NT, bool WIDE>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_gather(bb_gather_args a)
{
    constexpr int NCODE = 1 << BPS;
    constexpr uint32_t CMASK = NCODE - 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    // layout: [nslot][pitch] dwords of raw payload (one 256-byte block more
    // than the group needs: loads are whole aligned blocks, see
    // k_decode_flat_aln), then per slot a validity word and the byte offset of
    // the group's first payload dword inside its row, then the level table
    const uint32_t pitch = a.gtiles * 64 + 64 + 1;      // +1 dword: bank skew
    uint32_t *s_raw = s_mem;
    uint32_t *s_valid = s_mem + (size_t)a.nslot * pitch;
    uint32_t *s_base = s_valid + a.nslot;
    float *s_tab = reinterpret_cast<float *>(s_base + a.nslot + 1);

    // (levels as plain locals, not the bb_levels struct: captured by reference
    // by the lambda below, the struct was kept on the stack in the table
    // variants -- 16 bytes of scratch per lane)
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (LV == BB_LV_LDS) {
        for (int i = threadIdx.x; i < NCODE; i += BB_BLOCK) s_tab[i] = a.tab[i];
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
    const uint64_t E = a.ndw * (32 / BPS);              // elements per slot
    const uint64_t R = E >> a.lchunk;                   // rows per frame set
    const uint32_t rowlen = a.nslot << a.lchunk;        // floats per output row
    const uint64_t nwork = a.nframes * a.ngroup;
    const uint32_t gdw = a.gtiles * 64;                 // dwords staged per slot

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t f = work / a.ngroup;
        const uint32_t g = (uint32_t)(work - f * a.ngroup);
        const uint64_t dw0 = (uint64_t)g * gdw;         // first dword of the group
        // phase 1: stage the group's dwords of every slot (one wave per
        // slot at a time: coalesced 256-byte loads, no index arithmetic)
        if (threadIdx.x == 0) s_base[a.nslot] = 0;      // count of missing slots
        __syncthreads();
        bb_gather_stage(a, f, dw0, gdw, pitch, s_raw, s_valid, s_base, &s_base[a.nslot]);
        __syncthreads();
        const bool holes = s_base[a.nslot] != 0;        // uniform
        // phase 2: contiguous output region of this group
        const uint64_t e_lo = dw0 * (32 / BPS);
        const uint64_t e_hi = (e_lo + (uint64_t)gdw * (32 / BPS) < E) ? e_lo + (uint64_t)gdw * (32 / BPS) : E;
        const uint32_t nrow = (uint32_t)((e_hi - e_lo) >> a.lchunk);
        const uint32_t nfloat = nrow * rowlen;          // multiple of 4
        float *obase = a.out + (f * R + (e_lo >> a.lchunk)) * rowlen;
        const uint8_t *rawb = reinterpret_cast<const uint8_t *>(s_raw);
        if (!WIDE && a.lrow >= 2 && rowlen <= BB_BLOCK * 4) {
            // Narrow chunks with a power-of-two row of at most 1024 floats (8
            // threads x 1 channel: 8): a lane's float4 sits at the SAME place
            // of its row in every pass (the workgroup advances by 1024 floats),
            // so which slot and which position of the thread sample each of its
            // four floats comes from -- and that slot's LDS base -- are looked up
            // once per work item instead of once per float.
            const uint32_t rem0 = (threadIdx.x * 4) & (rowlen - 1);
            uint32_t wbit[4], base[4];
            bool hole[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t rem = rem0 + j;
                const uint32_t sj = rem >> a.lchunk;
                wbit[j] = (rem & (a.chunk - 1)) * BPS;
                base[j] = s_base[sj];
                hole[j] = holes && !s_valid[sj];
            }
            for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
                const uint32_t rb = ((q >> a.lrow) << a.lchunk) * BPS;
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t bit = rb + wbit[j];
                    const uint32_t code = ((uint32_t)rawb[base[j] + (bit >> 3)] >> (bit & 7)) & CMASK;
                    r[j] = level(code);
                    if (hole[j]) r[j] = (a.complex_data && ((wbit[j] / BPS) & 1)) ? a.fill_im : a.fill_re;
                }
                bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
            }
            __syncthreads();
            continue;
        }
        if (WIDE && a.lrow >= 2 && rowlen <= BB_BLOCK * 4) {
            // the same for chunks of four floats and more: the lane's slot and
            // its position inside the thread sample do not change
            const uint32_t rem0 = (threadIdx.x * 4) & (rowlen - 1);
            const uint32_t sj = rem0 >> a.lchunk;
            const uint32_t wbit = (rem0 & (a.chunk - 1)) * BPS;
            const uint32_t base = s_base[sj];
            const bool hole = holes && !s_valid[sj];
            for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
                const uint32_t bit = ((q >> a.lrow) << a.lchunk) * BPS + wbit;
                const uint32_t off = base + (bit >> 3);
                uint32_t bits;
                if (BPS == 8)      bits = *reinterpret_cast<const uint32_t *>(rawb + off);
                else if (BPS == 4) bits = *reinterpret_cast<const uint16_t *>(rawb + off);
                else               bits = (uint32_t)rawb[off] >> (bit & 7);
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r[j] = level((bits >> (j * BPS)) & CMASK);
                    if (hole) r[j] = (a.complex_data && (j & 1)) ? a.fill_im : a.fill_re;
                }
                bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
            }
            __syncthreads();
            continue;
        }
        for (uint32_t q = threadIdx.x * 4; q < nfloat; q += BB_BLOCK * 4) {
            uint32_t row, rem;
            if (a.lrow >= 0) { row = q >> a.lrow; rem = q & (rowlen - 1); }   // power-of-two rows
            else             { row = q / rowlen;  rem = q - row * rowlen; }
            float r[4];
            if (WIDE) {
                // the four elements belong to one thread slot and are adjacent
                // in its payload: one LDS read of 4 * BPS bits
                const uint32_t s = rem >> a.lchunk;
                const uint32_t within = rem & (a.chunk - 1);
                const uint32_t bit = ((row << a.lchunk) + within) * BPS;
                const uint32_t off = s_base[s] + (bit >> 3);
                uint32_t bits;
                if (BPS == 8)      bits = *reinterpret_cast<const uint32_t *>(rawb + off);
                else if (BPS == 4) bits = *reinterpret_cast<const uint16_t *>(rawb + off);
                else               bits = (uint32_t)rawb[off] >> (bit & 7);
                const bool hole = holes && !s_valid[s];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r[j] = level((bits >> (j * BPS)) & CMASK);
                    if (hole) r[j] = (a.complex_data && (j & 1)) ? a.fill_im : a.fill_re;
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t s = rem >> a.lchunk;
                const uint32_t within = rem & (a.chunk - 1);
                const uint32_t bit = ((row << a.lchunk) + within) * BPS;
                const uint32_t byte = rawb[s_base[s] + (bit >> 3)];
                const uint32_t code = (byte >> (bit & 7)) & CMASK;
                r[j] = level(code);
                if (holes && !s_valid[s])
                    r[j] = (a.complex_data && (within & 1)) ? a.fill_im : a.fill_re;
                if (++rem == rowlen) { rem = 0; ++row; }
            }
            }
            bb_store4<NT>(obase + q, bb_f4{r[0], r[1], r[2], r[3]});
        }
        __syncthreads();
    }
}

