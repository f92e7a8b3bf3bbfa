// Flat LUT decode with an explicit, launch-size independent write front.
//
// Same arithmetic as k_decode_flat_aln (k_flat.h: the reference's LUT take,
// vdif/payload.py:69-103, mark5b/payload.py:78-94, base/encoding.py:131-144,
// dada/payload.py:13-14, gsb/payload.py:24-36; fill base/frame.py:191-199) --
// what changes is WHICH piece of the output a wave writes WHEN.
//
// k_decode_flat_aln maps work item -> workgroup as blockIdx + k * gridDim with a
// grid of ~10^5 workgroups: the ~2000 resident workgroups then write into
// min(16, items per workgroup) windows that lie gridDim items apart, and how
// many windows there are depends on the launch size (docs/DESIGN_rounds1-3.md section 3,
// "Launch size": 5.1-5.6 TB/s for 9-70 GB launches against 6.4-6.6 for 145 GB).
// Here the map is explicit.  Work items (TPW tiles of 256 input bytes = up to
// TPW x 4 KiB of output for 2-bit data) are numbered in output order.  The
// grid is cut into GROUPS of G consecutive workgroups (G ~ the number of
// workgroups resident on the chip); a group owns a contiguous region of
// K * G * NW items and sweeps it in K steps: at step k, wave w of the group's
// j-th workgroup decodes item  region + k * (G * NW) + j * NW + w.  All waves
// of a group thus write side by side into one moving front of G * NW items
// (tens of MiB) -- the store pattern of a grid-stride fill, which is what the
// HBM system sustains best (tools/kbench: 6.9-7.3 TB/s) -- whatever the launch
// size.  Groups follow each other in dispatch order.  Input loads for step
// k + 1 are issued before the stores of step k (register double buffer), as
// 256-byte ALIGNED blocks with the payload's misalignment folded into the bit
// hand-out (see k_decode_flat_aln).
#pragma once
#include "k_flat.h"

struct bb_front_geom {
    uint64_t nitems;        // nfs * ipf
    uint32_t ipf;           // work items per frame-slot = ceil(ntiles / TPW)
    uint32_t G, K;          // workgroups per group, steps per group
    uint32_t step_fs;       // (G * NW) / ipf   } one step advances an item index
    uint32_t step_j;        // (G * NW) % ipf   } by G * NW
};

template <int BPS, int LV, bool NT, int NW, int TPW>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_flat_front(bb_flat_args a, bb_front_geom g)
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
    const int wave = __builtin_amdgcn_readfirstlane(bb_wave());
    const uint64_t E = a.ndw * (32 / BPS);
    const uint32_t ntiles = (uint32_t)((a.ndw + 63) / 64);
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

    // first item of this wave, as (frame-slot, item within it); wave-uniform
    const uint32_t grp = blockIdx.x / g.G, jb = blockIdx.x - grp * g.G;
    const uint64_t ws = (uint64_t)g.G * NW;                     // items per step
    uint64_t item = (uint64_t)grp * g.K * ws + (uint64_t)jb * NW + wave;
    uint64_t fs = item / g.ipf;
    uint32_t j = (uint32_t)(item - fs * g.ipf);

    uint32_t cur[TPW + 1], nxt[TPW + 1];
    bool cur_valid = false, nxt_valid = false;
    uint32_t cur_s = 0, nxt_s = 0;

    auto issue = [&](uint64_t f, uint32_t jj, uint32_t (&w)[TPW + 1], bool &valid, uint32_t &s) {
        const int64_t so = a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride;
        valid = so >= 0;
        const uint8_t *p = a.buf + (valid ? so : 0);
        // misalignment of the payload against 256-byte blocks of the address
        // space; payloads at odd byte offsets (repaired files) keep plain loads
        const uintptr_t ad = reinterpret_cast<uintptr_t>(p);
        s = (ad & 3) ? 0u : (uint32_t)((ad >> 2) & 63);
        const uint32_t *blk = reinterpret_cast<const uint32_t *>(p) - s;
        const uint32_t tile0 = jj * TPW;
#pragma unroll
        for (int u = 0; u <= TPW; ++u) {
            const uint64_t q = (uint64_t)(tile0 + u) * 64 + lane;   // block dword q = payload dword q - s
            const bool want = valid && q >= s && q - s < a.ndw && (u < TPW || s != 0);
            w[u] = want ? blk[q] : 0u;
        }
    };

    if (item < g.nitems) issue(fs, j, cur, cur_valid, cur_s);
    for (uint32_t k = 0; k < g.K && item < g.nitems; ++k) {
        // position of the next step
        uint64_t nfs_ = fs + g.step_fs;
        uint32_t nj = j + g.step_j;
        if (nj >= g.ipf) { nj -= g.ipf; ++nfs_; }
        const uint64_t nitem = item + ws;
        if (k + 1 < g.K && nitem < g.nitems) issue(nfs_, nj, nxt, nxt_valid, nxt_s);

        float *obase = a.out + fs * E;
        const uint32_t tile0 = j * TPW;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint32_t tile = tile0 + u;
            const bool live = tile < ntiles;                    // wave-uniform
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const uint32_t idx = (uint32_t)(BPS == 8 ? lane : p * 8 * BPS + src_lane0) + cur_s;
                const uint32_t lo = (uint32_t)__shfl((int)cur[u], (int)(idx & 63));
                uint32_t bits = lo;
                if (cur_s) {                                    // uniform: aligned frames need one shuffle
                    const uint32_t hi = (uint32_t)__shfl((int)cur[u + 1], (int)(idx & 63));
                    bits = idx >= 64 ? hi : lo;
                }
                bits >>= (BPS == 8 ? 0 : shift);
                const uint64_t e0 = (uint64_t)tile * EPT + 256 * p + 4 * lane;
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
                bb_store4<NT>(obase + e0, v);
            }
        }
#pragma unroll
        for (int u = 0; u <= TPW; ++u) cur[u] = nxt[u];
        cur_valid = nxt_valid;
        cur_s = nxt_s;
        fs = nfs_; j = nj; item = nitem;
    }
}

// ---- one pass, striped ("elementwise") form ---------------------------------
// tools/kbench.cpp, KB_ELEM (profiles/r02s_kbench_elem.log): a 2-bit decode
// written like a fill -- every thread owns a few float4 of the output, loads
// their input first, then stores; the grid is sized for ONE pass; a thread's
// pieces lie in different stripes of the output, so there are a few write
// fronts that advance with the dispatch order -- runs at 6.25-6.33 TB/s for 8.5,
// 34 and 137 GB of output alike, where the workgroup-per-frame and persistent
// kernels run at 5.4-5.65 below ~33 GB (one physical region, docs/DESIGN_rounds1-3.md 3.2).
// This is that form for real frames: a wave owns U tiles (256 input bytes ->
// 8/BPS KiB of output each), tile u in stripe u of the launch's tile sequence;
// all U loads are issued before the first store.  One pass: grid = tiles per
// stripe / waves per workgroup.
template <int BPS, int LV, bool NT, int U>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_flat_es(bb_flat_args a, uint64_t ntile_total, uint64_t per_stripe)
{
    constexpr int NCODE = 1 << BPS;
    constexpr int EPT = 2048 / BPS;
    constexpr int PASSES = 8 / BPS;
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
    const uint64_t g = (uint64_t)blockIdx.x * BB_WAVES_PER_BLOCK + (uint64_t)__builtin_amdgcn_readfirstlane(bb_wave());
    if (g >= per_stripe) return;                            // wave-uniform
    const uint64_t E = a.ndw * (32 / BPS);
    const uint32_t ntiles = (uint32_t)((a.ndw + 63) / 64);
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

    uint32_t w[U];
    uint64_t fsv[U];
    uint32_t tilev[U];
    bool validv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t t = (uint64_t)u * per_stripe + g;
        w[u] = 0u; fsv[u] = ~0ull; tilev[u] = 0; validv[u] = false;
        if (t < ntile_total) {                              // wave-uniform
            const uint64_t fs = t / ntiles;
            const uint32_t tile = (uint32_t)(t - fs * ntiles);
            const int64_t so = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
            fsv[u] = fs; tilev[u] = tile; validv[u] = so >= 0;
            const uint64_t dw = (uint64_t)tile * 64 + lane;
            if (so >= 0 && dw < a.ndw)
                w[u] = reinterpret_cast<const uint32_t *>(a.buf + so)[dw];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (fsv[u] == ~0ull) continue;
        float *obase = a.out + fsv[u] * E;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            uint32_t bits;
            if (BPS == 8) bits = w[u];
            else bits = (uint32_t)__shfl((int)w[u], p * 8 * BPS + src_lane0) >> shift;
            const uint64_t e0 = (uint64_t)tilev[u] * EPT + 256 * p + 4 * lane;
            if (e0 >= E) continue;
            bb_f4 v;
            if (validv[u]) {
                v.x = lv.get(bits & CMASK);
                v.y = lv.get((bits >> BPS) & CMASK);
                v.z = lv.get((bits >> (2 * BPS)) & CMASK);
                v.w = lv.get((bits >> (3 * BPS)) & CMASK);
            } else {
                v = fillv;
            }
            bb_store4<NT>(obase + e0, v);
        }
    }
}

// ---- the fill-shaped decode for real frames ---------------------------------
// tools/kbench.cpp KB_ELEM carried over with the frame arithmetic it needs: ONE
// pass; thread j of the grid owns float4 number j of each of the S = 16 stripes
// of the launch (a stripe = frames [k * fps, (k + 1) * fps)), i.e. 16 float4
// that lie fps frames apart; their 16 input dwords are loaded first (every
// stripe has the same frame geometry, so ONE division per thread places all of
// them), then come the 16 stores.  A wave-store is 1 KiB contiguous and
// consecutive workgroups write consecutive 4 KiB of every stripe: 16 write
// fronts that advance with the dispatch order, like a fill's.
#define BB_ELEM_STRIPES 16
template <int BPS, int LV, bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_flat_elem(bb_flat_args a, uint64_t fps)
{
    constexpr int NCODE = 1 << BPS;
    constexpr uint32_t CMASK = NCODE - 1;
    constexpr uint32_t FPD = 8 / BPS;                       // float4 per input dword
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
    const uint64_t E = a.ndw * (32 / BPS);                  // floats per frame-slot (a multiple of 4)
    const uint32_t E4 = (uint32_t)(E >> 2);                 // float4 per frame-slot
    const uint64_t j = (uint64_t)blockIdx.x * BB_BLOCK + threadIdx.x;
    // position inside a stripe -> (frame of the stripe, float4 of the frame)
    uint64_t fl;
    uint32_t r;
    if ((fps * E4) >> 32) { fl = j / E4; r = (uint32_t)(j - fl * E4); }
    else { const uint32_t q = (uint32_t)j / E4; fl = q; r = (uint32_t)j - q * E4; }
    if (fl >= fps) return;
    const uint32_t dwi = r / FPD;                           // input dword of the payload (FPD is a power of two)
    const uint32_t sh = (r & (FPD - 1)) * 4 * BPS;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};

    uint32_t w[BB_ELEM_STRIPES];
    int64_t sov[BB_ELEM_STRIPES];
#pragma unroll
    for (int k = 0; k < BB_ELEM_STRIPES; ++k) {
        const uint64_t f = (uint64_t)k * fps + fl;
        sov[k] = f < a.nfs ? (a.src ? a.src[f] : a.src0 + (int64_t)f * a.src_stride) : -2;
    }
#pragma unroll
    for (int k = 0; k < BB_ELEM_STRIPES; ++k) {
        w[k] = 0u;
        if (sov[k] >= 0) {
            const uint8_t *p = a.buf + sov[k] + (uint64_t)dwi * 4;
            if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) w[k] = *reinterpret_cast<const uint32_t *>(p);
            else w[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        }
    }
#pragma unroll
    for (int k = 0; k < BB_ELEM_STRIPES; ++k) {
        if (sov[k] == -2) continue;                         // beyond the last frame
        const uint64_t f = (uint64_t)k * fps + fl;
        bb_f4 v;
        if (sov[k] >= 0) {
            const uint32_t bits = w[k] >> sh;
            v.x = lv.get(bits & CMASK);
            v.y = lv.get((bits >> BPS) & CMASK);
            v.z = lv.get((bits >> (2 * BPS)) & CMASK);
            v.w = lv.get((bits >> (3 * BPS)) & CMASK);
        } else {
            v = fillv;
        }
        bb_store4<NT>(a.out + f * E + 4ull * r, v);
    }
}
