// Flat LUT decode: packed 1/2/4/8-bit codes -> float32, whole frames.
//
// Replaces (reference, path:line): lut.take(words.view(u1), axis=0)
// (vdif/payload.py:69-103, mark5b/payload.py:78-94), decode_8bit
// (base/encoding.py:131-144), int8.astype(f32) (dada/payload.py:13-14,
// gsb/payload.py:39-42), the sign-extending nibble split of
// gsb/payload.py:24-36, PayloadBase._decode (base/payload.py:314-330), the
// frameset thread interleave (vdif/frame.py:402-434) and the invalid-frame
// fill (base/frame.py:191-199).
//
// Shape (gfx950): one workgroup (4 waves) per frame-slot segment.  A wave
// reads a 256-byte tile with one coalesced dword-per-lane load, then emits
// 8/bps store passes.  In every pass lane l writes the float4 holding
// elements [256p + 4l, 256p + 4l + 4) of the tile, so each store instruction
// covers 1 KiB of contiguous output (8 full 128-byte lines).  The packed bits
// a lane needs sit in another lane's dword; they are fetched with one
// ds_bpermute (__shfl) per pass -- no LDS allocation, no barrier.
#pragma once
#include "bb_common.h"

enum { BB_OUT_FLAT = 0,     // nslot == 1: frame output is contiguous
       BB_OUT_ROWS4 = 1,    // nslot > 1, chunk % 4 == 0: float4 stays in a row chunk
       BB_OUT_SCATTER = 2 };// nslot > 1, chunk < 4: scalar stores (correctness path)

enum { BB_LV_REG = 0,       // 2 or 4 levels held in registers (bps 1, 2)
       BB_LV_LDS = 1,       // 16 / 256 entry table in LDS (bps 4; VDIF bps 8)
       BB_LV_INT8 = 2 };    // int8 -> f32 conversion (no table)

#define BB_SEG_TILES 32     // tiles (of 256 input bytes) per workgroup work item
#define BB_FLAT_UNROLL 4

struct bb_flat_args {
    const uint8_t *buf;
    const int64_t *src;     // [nfs] payload offsets, -1 = fill; may be null
    float         *out;
    const float   *tab;     // g_levels[coder][log2 bps]
    uint64_t nfs;           // nframes * nslot
    uint64_t ndw;           // payload dwords per frame-slot
    uint64_t nseg;          // work items per frame-slot
    uint32_t seg_tiles;     // tiles per work item (<= BB_SEG_TILES), evenly split
    uint32_t tpw;           // tiles per wave within a work item (<= 8)
    int64_t  src0, src_stride;
    uint32_t nslot, chunk, lchunk;
    float    fill_re, fill_im;
    int32_t  complex_data;
    uint64_t src_lim;       // offsets so with (uint64_t)so >= src_lim decode as fill (bb_src_ok)
    bb_perm_t perm;         // work order (bb_common.h)
#if BB_EXP
    int32_t  nt_loads;      // experiment: non-temporal input loads
    uint64_t *trace;        // experiment: completion time (wall_clock64) per work item, or null
    // experiment (BB_TUNE_OUT_STRIPE_*): frame-slot fs is written to output slot
    // (fs % stripe_w) * stripe_s + fs / stripe_w instead of slot fs, i.e. the
    // launch's output is dealt over stripe_w regions that lie stripe_s slots
    // apart; 0 = off.  Contiguous-output kernels (k_decode_flat, _aln) only.
    uint32_t stripe_w;
    uint64_t stripe_s;
#endif
};

__device__ __forceinline__ uint64_t bb_out_slot(const bb_flat_args &a, uint64_t fs)
{
#if BB_EXP
    if (a.stripe_w == 0) return fs;
    const uint64_t q = fs / a.stripe_w;
    return (fs - q * a.stripe_w) * a.stripe_s + q;
#else
    return fs;
#endif
}

// input dword load (the experiment build can ask for the non-temporal form)
__device__ __forceinline__ uint32_t bb_load_dw(const bb_flat_args &a, const uint32_t *p)
{
#if BB_EXP
    if (a.nt_loads) return __builtin_nontemporal_load(p);
#endif
    return *p;
}

template <int BPS, int LV>
struct bb_levels {
    float t0, t1, t2, t3;
    const float *lds;
    __device__ __forceinline__ float get(uint32_t code) const {
        if (LV == BB_LV_REG) {
            if (BPS == 1) return code ? t1 : t0;
            const float lo = (code & 1) ? t1 : t0;
            const float hi = (code & 1) ? t3 : t2;
            return (code & 2) ? hi : lo;
        } else if (LV == BB_LV_LDS) {
            return lds[code];
        } else {
            return (float)(int)(int8_t)code;
        }
    }
};

template <int BPS, int LV, int OM, bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_flat(bb_flat_args a)
{
    constexpr int NCODE = 1 << BPS;
    constexpr int EPT = 2048 / BPS;         // elements per 256-byte tile
    constexpr int PASSES = 8 / BPS;         // store passes per tile
    constexpr uint32_t CMASK = NCODE - 1;
    __shared__ float s_tab[LV == BB_LV_LDS ? NCODE : 1];

    bb_levels<BPS, LV> lv;
    lv.lds = s_tab;
    if (LV == BB_LV_LDS) {
        for (int i = threadIdx.x; i < NCODE; i += BB_BLOCK) s_tab[i] = a.tab[i];
        __syncthreads();
    } else if (LV == BB_LV_REG) {
        lv.t0 = a.tab[0]; lv.t1 = a.tab[1];
        if (BPS == 2) { lv.t2 = a.tab[2]; lv.t3 = a.tab[3]; }
    }

    const int lane = bb_lane();
    const int wave = bb_wave();
    const uint64_t E = a.ndw * (32 / BPS);          // elements per frame-slot
    const uint64_t ntiles = (a.ndw + 63) / 64;
    const uint64_t nwork = a.nfs * a.nseg;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    // lane-constant shuffle geometry
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

    for (uint64_t step = blockIdx.x; step < nwork; step += gridDim.x) {
        const uint64_t work = bb_perm(a.perm, step);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = work; seg = 0; }
        else { fs = work / a.nseg; seg = work - fs * a.nseg; }
        const int64_t so = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint32_t *in = reinterpret_cast<const uint32_t *>(a.buf + (valid ? so : 0));

        float *obase;           // FLAT: start of this frame-slot's output
        uint64_t rowbase = 0, slot = 0;
        if (OM == BB_OUT_FLAT) {
            obase = a.out + bb_out_slot(a, fs) * E;
        } else {
            const uint64_t f = fs / a.nslot;
            slot = fs - f * a.nslot;
            rowbase = f * (E >> a.lchunk);
            obase = a.out;
        }

        const uint64_t tile_begin = seg * a.seg_tiles;
        const uint64_t tile_end = (tile_begin + a.seg_tiles < ntiles)
                                  ? tile_begin + a.seg_tiles : ntiles;
        for (uint64_t t = tile_begin + wave; t < tile_end;
             t += BB_WAVES_PER_BLOCK * BB_FLAT_UNROLL) {
            uint32_t w[BB_FLAT_UNROLL];
#pragma unroll
            for (int u = 0; u < BB_FLAT_UNROLL; ++u) {
                const uint64_t tile = t + (uint64_t)u * BB_WAVES_PER_BLOCK;
                const uint64_t dw = tile * 64 + lane;
                w[u] = (valid && tile < tile_end && dw < a.ndw) ? in[dw] : 0u;
            }
#pragma unroll
            for (int u = 0; u < BB_FLAT_UNROLL; ++u) {
                const uint64_t tile = t + (uint64_t)u * BB_WAVES_PER_BLOCK;
                if (tile >= tile_end) break;            // wave-uniform
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    uint32_t bits;
                    if (BPS == 8) bits = w[u];
                    else bits = (uint32_t)__shfl((int)w[u], p * 8 * BPS + src_lane0) >> shift;
                    const uint64_t e0 = tile * EPT + 256 * p + 4 * lane;
                    if (e0 >= E) continue;
                    bb_f4 v;
                    if (valid) {
                        v.x = lv.get(bits & CMASK);
                        v.y = lv.get((bits >> BPS) & CMASK);
                        v.z = lv.get((bits >> (2 * BPS)) & CMASK);
                        v.w = lv.get((bits >> (3 * BPS)) & CMASK);
                    } else {
                        v = fillv;
                    }
                    if (OM == BB_OUT_FLAT) {
                        bb_store4<NT>(obase + e0, v);
                    } else if (OM == BB_OUT_ROWS4) {
                        const uint64_t row = e0 >> a.lchunk;
                        const uint64_t within = e0 & (a.chunk - 1);
                        bb_store4<NT>(obase + ((((rowbase + row) * a.nslot + slot) << a.lchunk) + within), v);
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint64_t e = e0 + j;
                            const uint64_t row = e >> a.lchunk;
                            const uint64_t within = e & (a.chunk - 1);
                            bb_store1<NT>(obase + ((((rowbase + row) * a.nslot + slot) << a.lchunk) + within), vv[j]);
                        }
                    }
                }
            }
        }
#if BB_EXP
        if (a.trace && threadIdx.x == 0) a.trace[work] = wall_clock64();
#endif
    }
}

// Thread-interleaved output with chunks of >= 4 floats (e.g. 8 threads x 16
// complex channels: 128-byte chunks in 1 KiB rows).  One wave per thread
// slot; all waves of a workgroup walk the SAME tiles of their payloads, so
// the rows of the output region are completed by the workgroup within a few
// microseconds instead of being visited by eight workgroups at different
// times.  Persistent and software pipelined like k_decode_flat_pipe.
// ALN: 256-byte aligned block loads with the misalignment folded into the
// bit hand-out, as in k_decode_flat_aln (every wave has its own payload offset).
template <int BPS, int LV, bool NT, int NW, int TPW, bool ALN>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_rows_pipe(bb_flat_args a)
{
    constexpr int NCODE = 1 << BPS;
    constexpr int EPT = 2048 / BPS;
    constexpr int PASSES = 8 / BPS;
    constexpr uint32_t CMASK = NCODE - 1;
    __shared__ float s_tab[LV == BB_LV_LDS ? NCODE : 1];
    bb_levels<BPS, LV> lv;
    lv.lds = s_tab;
    if (LV == BB_LV_LDS) {
        for (int i = threadIdx.x; i < NCODE; i += NW * BB_WAVE) s_tab[i] = a.tab[i];
        __syncthreads();
    } else if (LV == BB_LV_REG) {
        lv.t0 = a.tab[0]; lv.t1 = a.tab[1];
        if (BPS == 2) { lv.t2 = a.tab[2]; lv.t3 = a.tab[3]; }
    }
    const int lane = bb_lane();
    const int wave = bb_wave();
    const uint64_t E = a.ndw * (32 / BPS);
    const uint64_t R = E >> a.lchunk;
    const uint64_t nframes = a.nfs / a.nslot;
    const uint32_t sgroups = (a.nslot + NW - 1) / NW;       // slot groups of NW waves
    const uint64_t nwork = nframes * a.nseg * sgroups;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

    constexpr int NR = TPW + (ALN ? 1 : 0);
    uint32_t cur[NR], nxt[NR];
    bool cur_valid = false, nxt_valid = false;
    uint32_t cur_s = 0, nxt_s = 0;

    // work -> (frame set, segment, slot group); slot = group * NW + wave
    auto split = [&](uint64_t step, uint64_t &f, uint64_t &seg, uint32_t &slot) {
        const uint64_t work = bb_perm(a.perm, step);
        const uint64_t per_f = a.nseg * sgroups;
        f = work / per_f;
        const uint64_t r = work - f * per_f;
        seg = r / sgroups;
        slot = (uint32_t)(r - seg * sgroups) * NW + wave;
    };
    auto issue = [&](uint64_t work, uint32_t (&w)[NR], bool &valid, uint32_t &s) {
        uint64_t f, seg; uint32_t slot;
        split(work, f, seg, slot);
        int64_t so = -1;
        if (slot < a.nslot)
            so = a.src ? a.src[f * a.nslot + slot]
                       : a.src0 + (int64_t)(f * a.nslot + slot) * a.src_stride;
        valid = bb_src_ok(so, a.src_lim);
        const uint64_t tile0 = seg * a.seg_tiles;
        if (ALN) {
            const uint8_t *pp = a.buf + (valid ? (uint64_t)so : 0);
            const uintptr_t b0 = reinterpret_cast<uintptr_t>(pp);   // odd byte addresses: s = 0
            s = (b0 & 3) ? 0u : (uint32_t)((b0 >> 2) & 63);
            const uint32_t *blk = reinterpret_cast<const uint32_t *>(pp) - s;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const uint64_t j = (tile0 + u) * 64 + lane;     // block dword j = payload dword j - s
                w[u] = (valid && u <= (int)a.seg_tiles && j >= s && j - s < a.ndw) ? blk[j] : 0u;
            }
        } else {
            s = 0;
            const uint32_t *in = reinterpret_cast<const uint32_t *>(a.buf + (valid ? so : 0));
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const uint64_t dw = (tile0 + u) * 64 + lane;
                w[u] = (valid && u < (int)a.seg_tiles && dw < a.ndw) ? in[dw] : 0u;
            }
        }
    };

    uint64_t work = blockIdx.x;
    if (work < nwork) issue(work, cur, cur_valid, cur_s);
    for (; work < nwork; work += gridDim.x) {
        const uint64_t next = work + gridDim.x;
        if (next < nwork) issue(next, nxt, nxt_valid, nxt_s);
        uint64_t f, seg; uint32_t slot;
        split(work, f, seg, slot);
        const uint64_t tile0 = seg * a.seg_tiles;
        const uint64_t rowbase = f * R;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t tile = tile0 + u;
            const bool live = u < (int)a.seg_tiles && slot < a.nslot;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                uint32_t bits;
                if (ALN) {
                    const uint32_t idx = (uint32_t)(BPS == 8 ? lane : p * 8 * BPS + src_lane0) + cur_s;
                    bits = (uint32_t)__shfl((int)cur[u], (int)(idx & 63));
                    if (cur_s) {                        // wave-uniform
                        const uint32_t hi = (uint32_t)__shfl((int)cur[u + (ALN ? 1 : 0)], (int)(idx & 63));
                        if (idx >= 64) bits = hi;
                    }
                    bits >>= (BPS == 8 ? 0 : shift);
                } else if (BPS == 8) {
                    bits = cur[u];
                } else {
                    bits = (uint32_t)__shfl((int)cur[u], p * 8 * BPS + src_lane0) >> shift;
                }
                const uint64_t e0 = tile * EPT + 256 * p + 4 * lane;
                if (!live || e0 >= E) continue;
                bb_f4 v;
                if (cur_valid) {
                    v.x = lv.get(bits & CMASK);
                    v.y = lv.get((bits >> BPS) & CMASK);
                    v.z = lv.get((bits >> (2 * BPS)) & CMASK);
                    v.w = lv.get((bits >> (3 * BPS)) & CMASK);
                } else {
                    v = fillv;
                }
                const uint64_t row = e0 >> a.lchunk;
                const uint64_t within = e0 & (a.chunk - 1);
                bb_store4<NT>(a.out + ((((rowbase + row) * a.nslot + slot) << a.lchunk) + within), v);
            }
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) cur[u] = nxt[u];
        cur_valid = nxt_valid;
        cur_s = nxt_s;
    }
}

