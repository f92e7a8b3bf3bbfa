// Experiment kernels (make EXPERIMENTS=1 -> libbbdecode_exp.so, used by tools/):
// the register-pipelined flat family that round 1 and 2 measured against --
// k_decode_flat_pipe, k_decode_flat_aln (round 1's headline kernel),
// k_decode_flat_span, k_decode_flat2_bytes -- and, through k_front.h, the
// write-front / one-pass experiments.  None of them is launched by the product
// library; they stay as the record of those measurements (docs/DESIGN_rounds1-3.md 3.2, 3.3).
#pragma once
#include "k_flat.h"

// Persistent, software-pipelined form of k_decode_flat: a fixed grid of
// workgroups walks the work items; before a workgroup emits the stores of its
// current item it has already issued the loads of its next one (register
// double buffer, 8 dwords per lane each), so the HBM read latency -- several
// microseconds while the write queues are saturated -- overlaps the store
// phase instead of preceding it.  Each wave owns 8 consecutive tiles, i.e. a
// contiguous 32 KiB run of the output.
template <int BPS, int LV, int OM, bool NT, int NW, int TPW>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_flat_pipe(bb_flat_args a)
{
    constexpr int NCODE = 1 << BPS;
    constexpr int EPT = 2048 / BPS;
    constexpr int PASSES = 8 / BPS;
    constexpr uint32_t CMASK = NCODE - 1;
    // NW waves per workgroup, TPW tiles per wave: a work item is at most
    // NW * TPW tiles; every wave owns a contiguous run of TPW tiles.
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
    const uint64_t nwork = a.nfs * a.nseg;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

    uint32_t cur[TPW], nxt[TPW];
    bool cur_valid = false, nxt_valid = false;

    auto issue = [&](uint64_t step, uint32_t (&w)[TPW], bool &valid) {
        const uint64_t work = bb_perm(a.perm, step);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = work; seg = 0; }
        else { fs = work / a.nseg; seg = work - fs * a.nseg; }
        const int64_t so = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
        valid = bb_src_ok(so, a.src_lim);
        const uint32_t *in = reinterpret_cast<const uint32_t *>(a.buf + (valid ? so : 0));
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t dw_end = (seg + 1) * a.seg_tiles * 64 < a.ndw
                                ? (seg + 1) * a.seg_tiles * 64 : a.ndw;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t dw = (tile0 + u) * 64 + lane;
            w[u] = (valid && u < (int)a.tpw && dw < dw_end)
                   ? bb_load_dw(a, &in[dw]) : 0u;
        }
    };

    uint64_t work = blockIdx.x;
    if (work < nwork) issue(work, cur, cur_valid);
    for (; work < nwork; work += gridDim.x) {
        const uint64_t next = work + gridDim.x;
        if (next < nwork) issue(next, nxt, nxt_valid);

        const uint64_t pwork = bb_perm(a.perm, work);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = pwork; seg = 0; }
        else { fs = pwork / a.nseg; seg = pwork - fs * a.nseg; }
        float *obase;
        uint64_t rowbase = 0, slot = 0;
        if (OM == BB_OUT_FLAT) {
            obase = a.out + fs * E;
        } else {
            const uint64_t f = fs / a.nslot;
            slot = fs - f * a.nslot;
            rowbase = f * (E >> a.lchunk);
            obase = a.out;
        }
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t seg_e_end = (seg + 1) * a.seg_tiles * EPT < E
                                   ? (seg + 1) * a.seg_tiles * EPT : E;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            // (no early exit: the loop must unroll completely so that cur[]
            // stays in registers; tiles beyond tpw fail the range test below)
            const uint64_t tile = tile0 + u;
            const bool live = u < (int)a.tpw;           // wave-uniform
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                uint32_t bits;
                if (BPS == 8) bits = cur[u];
                else bits = (uint32_t)__shfl((int)cur[u], p * 8 * BPS + src_lane0) >> shift;
                const uint64_t e0 = tile * EPT + 256 * p + 4 * lane;
                if (!live || e0 >= seg_e_end) continue;
                bb_f4 v;
                if (cur_valid) {
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
#pragma unroll
        for (int u = 0; u < TPW; ++u) cur[u] = nxt[u];
        cur_valid = nxt_valid;
    }
}

// k_decode_flat_pipe for contiguous output with 256-byte ALIGNED tile loads.
// A payload starts wherever its header ends (VDIF: 32 bytes into an 8032-byte
// frame), so the dword-per-lane tile loads of k_decode_flat_pipe straddle
// three 128-byte lines instead of covering two.  Here a wave loads the TPW+1
// aligned 256-byte blocks that cover its tiles; the misalignment s (dwords,
// wave-uniform per frame) is folded into the lane index of the bit hand-out:
// payload dword o of tile u is lane (o+s)&63 of block u or u+1.
template <int BPS, int LV, bool NT, int NW, int TPW>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_flat_aln(bb_flat_args a)
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
    const uint64_t nwork = a.nfs * a.nseg;
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

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
        // absolute dword index of the payload start in the buffer; its low six
        // bits are the misalignment against 256-byte blocks
        // payloads found by the byte-granular search may start at odd bytes:
        // those keep plain (hardware-unaligned) loads, s = 0
        // (taken from the ADDRESS, not the offset: the buffer may be a view
        // into a file image resident in HBM and start anywhere)
        const uint8_t *pp = a.buf + (valid ? (uint64_t)so : 0);
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(pp);
        s = (b0 & 3) ? 0u : (uint32_t)((b0 >> 2) & 63);
        const uint32_t *blk = reinterpret_cast<const uint32_t *>(pp) - s;
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t dw_end = (seg + 1) * a.seg_tiles * 64 < a.ndw
                                ? (seg + 1) * a.seg_tiles * 64 : a.ndw;
#pragma unroll
        for (int u = 0; u <= TPW; ++u) {
            // block dword j holds payload dword j - s
            const uint64_t j = (tile0 + u) * 64 + lane;
            const bool want = valid && u <= (int)a.tpw && j >= s && j - s < dw_end;
            w[u] = want ? bb_load_dw(a, &blk[j]) : 0u;
        }
    };

    uint64_t work = blockIdx.x;
    if (a.trace && threadIdx.x == 0) a.trace[nwork + blockIdx.x] = wall_clock64();   // workgroup start
    if (work < nwork) issue(work, cur, cur_valid, cur_s);
    for (; work < nwork; work += gridDim.x) {
        const uint64_t next = work + gridDim.x;
        if (next < nwork) issue(next, nxt, nxt_valid, nxt_s);

        const uint64_t pwork = bb_perm(a.perm, work);
        uint64_t fs, seg;
        if (a.nseg == 1) { fs = pwork; seg = 0; }
        else { fs = pwork / a.nseg; seg = pwork - fs * a.nseg; }
        float *obase = a.out + bb_out_slot(a, fs) * E;
        const uint64_t tile0 = seg * a.seg_tiles + (uint64_t)wave * a.tpw;
        const uint64_t seg_e_end = (seg + 1) * a.seg_tiles * EPT < E
                                   ? (seg + 1) * a.seg_tiles * EPT : E;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t tile = tile0 + u;
            const bool live = u < (int)a.tpw;           // wave-uniform
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const uint32_t idx = (uint32_t)(BPS == 8 ? lane : p * 8 * BPS + src_lane0) + cur_s;
                const uint32_t lo = (uint32_t)__shfl((int)cur[u], (int)(idx & 63));
                uint32_t bits = lo;
                if (cur_s) {                            // uniform: aligned frames need one shuffle
                    const uint32_t hi = (uint32_t)__shfl((int)cur[u + 1], (int)(idx & 63));
                    bits = idx >= 64 ? hi : lo;
                }
                bits >>= (BPS == 8 ? 0 : shift);
                const uint64_t e0 = tile * EPT + 256 * p + 4 * lane;
                if (!live || e0 >= seg_e_end) continue;
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
        if (a.trace && threadIdx.x == 0) a.trace[work] = wall_clock64();
#pragma unroll
        for (int u = 0; u <= TPW; ++u) cur[u] = nxt[u];
        cur_valid = nxt_valid;
        cur_s = nxt_s;
    }
}

// Frame-agnostic form of k_decode_flat_pipe for contiguous output (nslot == 1):
// the work is cut in OUTPUT space.  All payload dwords of the launch form one
// stream D = 0 .. nfs*ndw-1 (frame = D / ndw); a wave owns TPW consecutive
// 64-dword tiles of that stream, i.e. a run of the output that starts on a
// multiple of its own length (64 KiB for 2-bit data) no matter where frame
// boundaries fall.  With the per-frame split an 8000-byte payload is 31.25
// tiles: runs of 64 and 61 KiB at 1 KiB-aligned positions, which costs about
// 2.5 % of the HBM write rate against aligned runs
// (profiles/r01e_exp_frame_size.log).  Payloads are at least one run long
// (host side checks ndw >= 64 * TPW), so a run touches at most two frames, A
// and B: their payload offsets are wave-uniform, and a lane only decides on
// which side of the boundary its dword lies.  Frames without an entry (-1)
// decode to fill.
template <int BPS, int LV, bool NT, int NW, int TPW>
__global__ __launch_bounds__(NW * BB_WAVE)
void k_decode_flat_span(bb_flat_args a)
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
    const uint64_t W = a.ndw;                       // payload dwords per frame (>= 64 * TPW)
    const uint64_t dtot = a.nfs * W;
    const uint64_t etot = dtot * (32 / BPS);
    const uint64_t ntile = (dtot + 63) / 64;
    const uint64_t nwork = (ntile + NW * TPW - 1) / (NW * TPW);
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    const int src_lane0 = (lane * BPS) >> 3;
    const int shift = (4 * lane * BPS) & 31;

    auto offset_of = [&](uint64_t frame) -> int64_t {
        if (frame >= a.nfs) return -1;
        return a.src ? a.src[frame] : a.src0 + (int64_t)frame * a.src_stride;
    };

    struct span { uint64_t boundary; bool okA, okB; };   // wave-uniform
    uint32_t cur[TPW], nxt[TPW];
    span cs = {0, false, false}, ns = {0, false, false};

    auto issue = [&](uint64_t work, uint32_t (&w)[TPW], span &sp) {
        const uint64_t dstart = ((work * NW + wave) * TPW) * 64;      // uniform
        const uint64_t frameA = dstart / W;
        const int64_t soA = offset_of(frameA), soB = offset_of(frameA + 1);
        sp.boundary = (frameA + 1) * W;
        sp.okA = soA >= 0;
        sp.okB = soB >= 0;
        // dword pointers such that ptr[D] is stream dword D on either side
        const uint32_t *pa = reinterpret_cast<const uint32_t *>(a.buf + (sp.okA ? soA : 0)) - frameA * W;
        const uint32_t *pb = reinterpret_cast<const uint32_t *>(a.buf + (sp.okB ? soB : 0)) - sp.boundary;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t d = dstart + 64ull * u + lane;
            const bool inA = d < sp.boundary;
            const bool ok = d < dtot && (inA ? sp.okA : sp.okB);
            const uint32_t *q = (inA ? pa : pb) + d;
            w[u] = ok ? bb_load_dw(a, q) : 0u;
        }
    };

    uint64_t work = blockIdx.x;
    if (work < nwork) issue(work, cur, cs);
    for (; work < nwork; work += gridDim.x) {
        const uint64_t next = work + gridDim.x;
        if (next < nwork) issue(next, nxt, ns);

        const uint64_t tile0 = (work * NW + wave) * TPW;
        const bool holes = !(cs.okA && cs.okB);                 // uniform, rare
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const uint64_t tile = tile0 + u;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int owner = p * 8 * BPS + src_lane0;      // lane whose dword holds my bits
                uint32_t bits;
                if (BPS == 8) bits = cur[u];
                else bits = (uint32_t)__shfl((int)cur[u], owner) >> shift;
                const uint64_t e0 = tile * EPT + 256 * p + 4 * lane;
                if (e0 >= etot) continue;
                bb_f4 v;
                v.x = lv.get(bits & CMASK);
                v.y = lv.get((bits >> BPS) & CMASK);
                v.z = lv.get((bits >> (2 * BPS)) & CMASK);
                v.w = lv.get((bits >> (3 * BPS)) & CMASK);
                if (holes) {
                    const uint64_t d = tile * 64 + (BPS == 8 ? lane : owner);
                    if (!(d < cs.boundary ? cs.okA : cs.okB)) v = fillv;
                }
                bb_store4<NT>(a.out + e0, v);
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) cur[u] = nxt[u];
        cs = ns;
    }
}

// Experimental twin of the 2-bit flat kernel: every lane loads its own byte
// (64-byte wave loads) instead of shuffling a dword; kept for A/B timing only.
template <bool NT>
__global__ __launch_bounds__(BB_BLOCK)
void k_decode_flat2_bytes(bb_flat_args a)
{
    const float t0 = a.tab[0], t1 = a.tab[1], t2 = a.tab[2], t3 = a.tab[3];
    const int lane = bb_lane();
    const int wave = bb_wave();
    const uint64_t nbytes = a.ndw * 4;
    const uint64_t E = nbytes * 4;
    const uint64_t nunits = (nbytes + 63) / 64;     // 64 input bytes -> 1 KiB
    const bb_f4 fillv = a.complex_data
        ? bb_f4{a.fill_re, a.fill_im, a.fill_re, a.fill_im}
        : bb_f4{a.fill_re, a.fill_re, a.fill_re, a.fill_re};
    for (uint64_t fs = blockIdx.x; fs < a.nfs; fs += gridDim.x) {
        const int64_t so = a.src ? a.src[fs] : a.src0 + (int64_t)fs * a.src_stride;
        const bool valid = bb_src_ok(so, a.src_lim);
        const uint8_t *in = a.buf + (valid ? so : 0);
        float *obase = a.out + fs * E;
        for (uint64_t u0 = wave; u0 < nunits; u0 += BB_WAVES_PER_BLOCK * 8) {
            uint32_t b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint64_t i = (u0 + (uint64_t)k * BB_WAVES_PER_BLOCK) * 64 + lane;
                b[k] = (valid && i < nbytes) ? in[i] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint64_t i = (u0 + (uint64_t)k * BB_WAVES_PER_BLOCK) * 64 + lane;
                if (i >= nbytes) continue;
                bb_f4 v;
                if (valid) {
                    float r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t c = (b[k] >> (2 * j)) & 3;
                        const float lo = (c & 1) ? t1 : t0;
                        const float hi = (c & 1) ? t3 : t2;
                        r[j] = (c & 2) ? hi : lo;
                    }
                    v = bb_f4{r[0], r[1], r[2], r[3]};
                } else v = fillv;
                bb_store4<NT>(obase + i * 4, v);
            }
        }
    }
}

#include "k_front.h"
