// libbbdecode.so -- C ABI over the gfx950 kernels.  See include/bbdecode.h.
// Single translation unit: the kernel headers share the __device__ level
// tables defined in bb_common.h.
#include "bb_common.h"
#include "k_scan.h"
#include "k_flat.h"
#if BB_EXP
#include "k_exp.h"
#endif
#include "k_lut.h"
#include "k_lds.h"
#include "k_copy.h"
#include "k_tfpick.h"
#if BB_EXP
#include "k_burst.h"
#endif
#include "k_gather.h"
#include "k_pick.h"
#include "k_mark4.h"
#include "k_tiled.h"
#include "k_xpose.h"
#include "k_encode.h"

#include <atomic>
#include <mutex>
#include <stdio.h>
#include <string.h>
#include <type_traits>

namespace {

thread_local int t_last_hip = 0;
// name of the decode kernel the calling thread launched last (bb_last_kernel)
thread_local char t_last_kernel[160] = "";
#define BB_NOTE(...) snprintf(t_last_kernel, sizeof(t_last_kernel), __VA_ARGS__)
inline const char *lv_name(int bps, int coder)
{
    return bps <= 2 ? "REG" : (bps == 8 && coder == BB_CODER_INT) ? "INT8" : "LDS";
}

inline int hip_fail(hipError_t e) { t_last_hip = (int)e; return BB_EIO; }
#define BB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_); } while (0)

// ---- level tables ---------------------------------------------------------
float h_levels[3][4][256];
float h_enc2_thr[3];
std::once_flag h_levels_once;

// floats in increasing order <-> uint32 keys in increasing order
inline uint32_t float_key(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float key_float(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f; memcpy(&f, &u, 4);
    return f;
}

// Smallest float at which the reference 2-bit encoder reaches each code
// (the encoder is a monotone step function; see k_encode.h).
void fill_encode_thresholds()
{
    const uint32_t kmin = float_key(-3.4028234663852886e38f), kmax = float_key(3.4028234663852886e38f);
    for (uint32_t code = 1; code <= 3; ++code) {
        uint32_t lo = kmin, hi = kmax;            // code(lo) < code <= code(hi)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (bb_encode2_reference(key_float(mid)) >= code) hi = mid; else lo = mid;
        }
        h_enc2_thr[code - 1] = key_float(hi);
    }
}

void fill_host_levels()
{
    memset(h_levels, 0, sizeof(h_levels));
    fill_encode_thresholds();
    // base/encoding.py:14  OPTIMAL_2BIT_HIGH = 3.316505 -> float32
    const volatile float hi = 3.316505f;
    // VDIF, offset binary (base/encoding.py:52-56; vdif/payload.py:53-63)
    h_levels[BB_CODER_VDIF][0][0] = -1.0f; h_levels[BB_CODER_VDIF][0][1] = 1.0f;
    h_levels[BB_CODER_VDIF][1][0] = -hi;   h_levels[BB_CODER_VDIF][1][1] = -1.0f;
    h_levels[BB_CODER_VDIF][1][2] = 1.0f;  h_levels[BB_CODER_VDIF][1][3] = hi;
    const volatile float four_bit_1_sigma = 2.95f;       // base/encoding.py:46
    for (int n = 0; n < 16; ++n) {
        volatile float x = (float)n;
        x = x - 8.0f;
        x = x / four_bit_1_sigma;                         // true IEEE division
        h_levels[BB_CODER_VDIF][2][n] = x;
    }
    const volatile float eight_bit_1_sigma = 35.5f;      // base/encoding.py:48
    for (int n = 0; n < 256; ++n) {                       // base/encoding.py:141-143
        volatile float x = (float)n;
        x = x - 127.5f;
        x = x / eight_bit_1_sigma;
        h_levels[BB_CODER_VDIF][3][n] = x;
    }
    // Mark 5B, sign/magnitude (mark5b/payload.py:60-66)
    h_levels[BB_CODER_MARK5B][0][0] = 1.0f; h_levels[BB_CODER_MARK5B][0][1] = -1.0f;
    h_levels[BB_CODER_MARK5B][1][0] = -hi;  h_levels[BB_CODER_MARK5B][1][1] = 1.0f;
    h_levels[BB_CODER_MARK5B][1][2] = -1.0f; h_levels[BB_CODER_MARK5B][1][3] = hi;
    // two's complement integers (gsb/payload.py:24-42; dada/payload.py:13-14)
    for (int n = 0; n < 16; ++n)  h_levels[BB_CODER_INT][2][n] = (float)(n < 8 ? n : n - 16);
    for (int n = 0; n < 256; ++n) h_levels[BB_CODER_INT][3][n] = (float)(int8_t)(uint8_t)n;
}

inline int log2_bps(int bps)
{
    switch (bps) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; default: return -1; }
}

bool coder_supported(int coder, int bps)
{
    if (log2_bps(bps) < 0) return false;
    switch (coder) {
        case BB_CODER_VDIF:   return true;
        case BB_CODER_MARK5B: return bps == 1 || bps == 2;
        case BB_CODER_INT:    return bps == 4 || bps == 8;
        default: return false;
    }
}

std::atomic<uint64_t> g_dev_inited{0};
std::mutex g_init_mutex;

int ensure_init()
{
    int dev = 0;
    BB_HIP(hipGetDevice(&dev));
    if (dev < 64 && (g_dev_inited.load(std::memory_order_acquire) >> dev) & 1) return BB_OK;
    std::lock_guard<std::mutex> lock(g_init_mutex);
    std::call_once(h_levels_once, fill_host_levels);
    BB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_levels), h_levels, sizeof(h_levels)));
    BB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_enc2_thr), h_enc2_thr, sizeof(h_enc2_thr)));
    if (dev < 64) g_dev_inited.fetch_or(1ull << dev, std::memory_order_release);
    return BB_OK;
}

int device_levels(int coder, int lb, const float **p)
{
    float *base = nullptr;
    BB_HIP(hipGetSymbolAddress((void **)&base, HIP_SYMBOL(g_levels)));
    *p = base + ((size_t)coder * 4 + lb) * 256;
    return BB_OK;
}

// Persistent grids: measured optimum for all decode kernels is about 10^5
// workgroups (profiles/r01f_exp_grid.log: flat 6.22 -> 6.41 TB/s, thread
// interleave 5.84 -> 6.39, Mark 4 6.26 -> 6.37 against 4-16 thousand; one
// workgroup per work item is 10-15 % slower).  Far more workgroups than the
// ~4000 resident ones lets the dispatcher even out the tail, while each still
// walks several items with its loads one item ahead.
#define BB_GRID_CAP 131072ull
#define BB_PICK_NW 2                                // waves per workgroup of k_decode_pick (each has its own work item)
#define BB_LOCATE_GRID 65536ull      // the byte-granular searches (bb_*_locate)

// ---- tuning (include/bbdecode_tune.h; the experiment build adds bbdecode_exp.h) ----
// The knobs are THREAD-LOCAL: bb_tune() changes the geometry of the calling
// host thread's later launches only, never another thread's (VERDICT r3 next
// 8: the library keeps no process-wide mutable state besides the level tables
// and the arenas a caller creates).  Results never depend on a knob.
struct bb_knob {
    int v;
    int load() const { return v; }
    bb_knob &operator=(int x) { v = x; return *this; }
};
thread_local bb_knob g_tune_blocks{0};
thread_local bb_knob g_tune_tile_elems{8192};
thread_local bb_knob g_tune_encode_direct{0};
thread_local bb_knob g_tune_gather_bytes{8192};
thread_local bb_knob g_tune_tiled_stage{1};
thread_local bb_knob g_tune_seg_tiles{0};    // plain kernel: tiles per workgroup; 0 = 32, or 16 for 8-bit samples
thread_local bb_knob g_tune_gather_chunks{32};   // chunks below this many floats go through k_decode_gather
thread_local bb_knob g_tune_mkbf_tc{32};
thread_local bb_knob g_tune_rows_tiles{8};          // tiles per work item of k_decode_rows_pipe (1..8)
thread_local bb_knob g_tune_lut_tpw{0};             // tiles per wave and work item of the byte-table kernels; 0 = by kernel
static int touch_mib_default()                      // BB_TOUCH_MIB in the environment: the default of the knob below, for whole-process A/Bs
{
    static const int v = [] { const char *e = getenv("BB_TOUCH_MIB"); char *end = nullptr; const long x = e && *e ? strtol(e, &end, 10) : 256; return (e && *e && end == e) || x < 0 || x > (1 << 20) ? 256 : (int)x; }();
    return v;
}
thread_local bb_knob g_tune_touch_mib{touch_mib_default()};   // a read window of at most this many MiB is read through once before its decode (0 = never)
thread_local bb_knob g_tune_vdif8_lds_gib{20};     // GiB of payload from which VDIF 8-bit frames take k_decode_flat_lds<8,LDS,glds>
thread_local bb_knob g_tune_select_pick{1};        // folded channel subsets go through k_decode_pick where it applies
thread_local bb_knob g_tune_pick_bytes{4096};      // payload bytes (all slots) a wave of k_decode_pick stages per item
thread_local bb_knob g_tune_select_bytes{16384};   // payload bytes k_decode_gather_select stages per work item
thread_local bb_knob g_tune_m4_tiles{BB_M4_TPW};   // 64-word tiles per wave and work item of the Mark 4 decode kernels (1..8)
thread_local bb_knob g_tune_m4_widen{1};     // 1: 16-/32-track Mark 4 words decoded as 64-bit super-words (m4_widen)
thread_local bb_knob g_tune_xpose_rows{0};   // k_decode_i8_xpose: output rows per tile, 128 or 64; 0 = by layout
#if BB_EXP
thread_local bb_knob g_tune_flat8_lds{0};    // experiment: 1 = contiguous 8-bit output through k_decode_flat_lds<8>
#endif
thread_local bb_knob g_tune_encode_runs{0};  // k_encode_flat: 256-quad runs per wave and step (1 or 2); 0 = by sample width
thread_local bb_knob g_tune_encode_lw{0};    // k_encode_flat: log2(stripes) the runs are dealt over (bb_perm_t); 0 = input order
thread_local bb_knob g_tune_xpose_tc{0};     // k_decode_i8_xpose: channels per tile, 64 / 32 / 16 / 8; 0 = by channel count
thread_local bb_knob g_tune_xpose_min_nc{8}; // k_decode_i8_xpose without a selection: from this many channels on
thread_local bb_knob g_tune_xpose{1};        // 1: aligned int8 transposes through k_decode_i8_xpose; 0: k_tiled.h only
thread_local bb_knob g_tune_gather_glds{-1}; // gather kernels stage with direct-to-LDS loads: 1 / 0; -1 = by kernel (selecting: yes, +4.9 %; whole decodes: no, -0.2..-0.8 %; profiles/r04q_exp_gather_glds.log)
thread_local bb_knob g_tune_order_lw{-1};    // work order: log2(stripes) a launch is dealt over (bb_perm_t); 0 = file order, -1 = by output size
#if BB_EXP
thread_local bb_knob g_tune_m4_lds{0};       // 1: 64-bit Mark 4 words staged in LDS by direct-to-LDS loads (k_decode_mark4_lds)
thread_local bb_knob g_tune_copy{0};         // k_copy_frames: loads per lane | non-temporal loads << 8 (0 = 4 | 1 << 8)
thread_local bb_knob g_tune_variant{5};      // 5 = the product dispatch; others: include/bbdecode_exp.h
thread_local bb_knob g_tune_burst{0};         // 1: contiguous 2-bit output through k_decode_flat_burst (k_burst.h)
thread_local bb_knob g_tune_burst_bytes{65536};   // ... LDS bytes per staging buffer
thread_local bb_knob g_tune_burst_period{0};  // ... loader time slot, wall-clock ticks (10 ns); 0 = none
thread_local bb_knob g_tune_burst_waves{15};  // ... store waves per workgroup (3, 7 or 15)
thread_local bb_knob g_tune_nt{1};
thread_local bb_knob g_tune_nt_loads{0};
thread_local bb_knob g_tune_tpw{12};
std::atomic<uint64_t *> g_trace{nullptr};     // (experiment build: a process-wide debugging aid)
thread_local bb_knob g_tune_lds_pad{0};
thread_local bb_knob g_tune_tpw8{12};   // 8-bit data, aligned kernel: > 16 selects the 32-tile instantiation
thread_local bb_knob g_tune_lut_small{0};           // 1: the 4-tile instantiation of k_decode_flat_lut when items allow (experiment: slower)
thread_local bb_knob g_tune_byte_lut{1};     // 1: 1-/2-bit contiguous decode through the byte table kernel (k_lut.h)
thread_local bb_knob g_tune_stripe_w{0};     // experiment: output striping (bb_flat_args::stripe_w)
thread_local bb_knob g_tune_stripe_s{0};     // ... distance between the stripes, in frame-slots
thread_local bb_knob g_tune_front_g{2048};  // k_decode_flat_front: workgroups per group (one write front)
thread_local bb_knob g_tune_front_k{16};    // k_decode_flat_front: steps a group sweeps
inline bool tune_nt() { return g_tune_nt.load() != 0; }
#else
inline bool tune_nt() { return true; }
#endif

// Non-temporal stores are what the product launches; the experiment build can
// switch to plain stores (BB_TUNE_NT_STORES) and then instantiates both forms.
template <class F>
inline void with_nt(bool nt, F &&f)
{
#if BB_EXP
    if (!nt) { f(std::false_type{}); return; }
#endif
    (void)nt;
    f(std::true_type{});
}

// offsets taken from an index are followed only while the whole unit of
// `span` bytes lies inside the buffer (bb_src_ok)
inline uint64_t src_limit(size_t buf_nbytes, uint64_t span)
{
    return buf_nbytes >= span ? (uint64_t)buf_nbytes - span + 1 : 0;
}

// Work order of a launch of `nwork` items writing `out_bytes` (bb_common.h,
// bb_perm_t); launches of fewer than 64 items per stripe keep file order.
// Rounds 2-5 dealt every launch over 16 stripes from 16 GiB of output on and over 4
// below (profiles/r02e_exp_order.log: 33 GB outputs +12-15 % against file order, 8 GB:
// 4 stripes +0-2 %, 16 stripes -0-3 %).  Round 6 measured the number of stripes again,
// WITHIN one process on the same buffers (across processes the placement of the output
// drowns it: +-15 %), for every kernel family and 1 - 137 GB of output
// (profiles/r06o_exp_stripes_sizes.log, r06p_exp_stripes_headline.log,
// r06r_formats_stripes_sweep.log, r06n_exp_stripes_placement.log):
//   contiguous output (k_decode_flat_lds / _lut / int8, Mark 5B): 8 stripes +2.0-5.3 % below
//     16 GiB (every size from 0.5 GB up, arena blocks and plain allocations; +6-7 % where
//     the output's memory is physically contiguous and decodes slowly), +0.6-2.9 % above,
//     +1.1-1.3 % at the headline size;
//   GUPPI / MKBF transposes (k_decode_i8_xpose & co.): +2.6-5.6 % at every size;
//   the LDS gather (thread interleave): +4.5 % at 1 GB, +-0.4 % from 4 GB on: 8 below 16 GiB, 16 above as before;
//   k_decode_rows_pipe: 4 stripes stay best below 16 GiB (8: -0.6 .. -1.9 %), 8 above (+2.7 %);
//   Mark 4, channel selections, copies: within +-1.2 % of the old rule: unchanged.
enum bb_order_family { BB_ORDER_FLAT, BB_ORDER_GATHER, BB_ORDER_ROWS, BB_ORDER_TILED, BB_ORDER_OTHER };

bb_perm_t make_perm(uint64_t nwork, uint64_t out_bytes, bb_order_family family = BB_ORDER_OTHER)
{
    bb_perm_t p = {0, 0, 0};
    int lw = g_tune_order_lw.load();
    if (lw < 0) {
        const bool large = out_bytes >= (16ull << 30);
        switch (family) {
            case BB_ORDER_FLAT: case BB_ORDER_TILED: lw = 3; break;
            case BB_ORDER_GATHER: lw = large ? 4 : 3; break;
            case BB_ORDER_ROWS: lw = large ? 3 : 2; break;
            default: lw = large ? 4 : 2; break;
        }
    }
    if (lw > 0 && (nwork >> lw) >= 64) {
        p.lw = (uint32_t)lw;
        p.stripe = nwork >> lw;
        p.n = p.stripe << lw;
    }
    return p;
}

// (bits per sample, coder) -> kernel template arguments <BPS, LV>
template <class F>
inline void with_levels(int bps, int coder, F &&f)
{
    using std::integral_constant;
    switch (bps) {
        case 1: f(integral_constant<int, 1>{}, integral_constant<int, BB_LV_REG>{}); break;
        case 2: f(integral_constant<int, 2>{}, integral_constant<int, BB_LV_REG>{}); break;
        case 4: f(integral_constant<int, 4>{}, integral_constant<int, BB_LV_LDS>{}); break;
        default:
            if (coder == BB_CODER_INT) f(integral_constant<int, 8>{}, integral_constant<int, BB_LV_INT8>{});
            else                       f(integral_constant<int, 8>{}, integral_constant<int, BB_LV_LDS>{});
            break;
    }
}

void launch_gather(int bps, int coder, bool nt, dim3 grid, size_t lds, hipStream_t st, const bb_gather_args &a)
{
    const bool wide = a.lchunk >= 2;
    with_levels(bps, coder, [&](auto B, auto L) {
        with_nt(nt, [&](auto NT) {
            constexpr int BPS = decltype(B)::value, LV = decltype(L)::value;
            constexpr bool N = decltype(NT)::value;
            if (wide) hipLaunchKernelGGL((k_decode_gather<BPS, LV, N, true>), grid, dim3(BB_BLOCK), lds, st, a);
            else      hipLaunchKernelGGL((k_decode_gather<BPS, LV, N, false>), grid, dim3(BB_BLOCK), lds, st, a);
        });
    });
}

void launch_gather_select(int bps, int coder, bool nt, bool v4, dim3 grid, size_t lds, hipStream_t st,
                          const bb_gather_args &a)
{
    with_levels(bps, coder, [&](auto B, auto L) {
        with_nt(nt, [&](auto NT) {
            constexpr int BPS = decltype(B)::value, LV = decltype(L)::value;
            constexpr bool N = decltype(NT)::value;
            if (v4) hipLaunchKernelGGL((k_decode_gather_select<BPS, LV, N, true>), grid, dim3(BB_BLOCK), lds, st, a);
            else    hipLaunchKernelGGL((k_decode_gather_select<BPS, LV, N, false>), grid, dim3(BB_BLOCK), lds, st, a);
        });
    });
}

void launch_pick(int bps, int coder, bool nt, dim3 grid, size_t lds, hipStream_t st, const bb_pick_args &a)
{
    with_levels(bps, coder, [&](auto B, auto L) {
        with_nt(nt, [&](auto NT) {
            constexpr int BPS = decltype(B)::value, LV = decltype(L)::value;
            constexpr bool N = decltype(NT)::value;
            hipLaunchKernelGGL((k_decode_pick<BPS, LV, N, BB_PICK_NW>), grid, dim3(BB_PICK_NW * BB_WAVE), lds, st, a);
        });
    });
}

// thread interleave with wide chunks: one wave per thread slot (2, 4 or 8 per
// workgroup), aligned block loads
void launch_rows_pipe(int bps, int coder, bool nt, int nw, dim3 grid, hipStream_t st, const bb_flat_args &a)
{
    with_levels(bps, coder, [&](auto B, auto L) {
        with_nt(nt, [&](auto NT) {
            constexpr int BPS = decltype(B)::value, LV = decltype(L)::value;
            constexpr bool N = decltype(NT)::value;
            if (nw == 8)      hipLaunchKernelGGL((k_decode_rows_pipe<BPS, LV, N, 8, 8, true>), grid, dim3(8 * BB_WAVE), 0, st, a);
            else if (nw == 4) hipLaunchKernelGGL((k_decode_rows_pipe<BPS, LV, N, 4, 8, true>), grid, dim3(4 * BB_WAVE), 0, st, a);
            else              hipLaunchKernelGGL((k_decode_rows_pipe<BPS, LV, N, 2, 8, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
        });
    });
}

// the plain kernel: 8-bit samples with contiguous output, and the general
// fallback (thread interleave without an index or with more thread slots
// than the gather kernels stage)
void launch_flat(int bps, int coder, int om, bool nt, dim3 grid, hipStream_t st, const bb_flat_args &a)
{
    with_levels(bps, coder, [&](auto B, auto L) {
        with_nt(nt, [&](auto NT) {
            constexpr int BPS = decltype(B)::value, LV = decltype(L)::value;
            constexpr bool N = decltype(NT)::value;
            if (om == BB_OUT_FLAT)       hipLaunchKernelGGL((k_decode_flat<BPS, LV, BB_OUT_FLAT, N>), grid, dim3(BB_BLOCK), 0, st, a);
            else if (om == BB_OUT_ROWS4) hipLaunchKernelGGL((k_decode_flat<BPS, LV, BB_OUT_ROWS4, N>), grid, dim3(BB_BLOCK), 0, st, a);
            else                         hipLaunchKernelGGL((k_decode_flat<BPS, LV, BB_OUT_SCATTER, N>), grid, dim3(BB_BLOCK), 0, st, a);
        });
    });
}

// (direct arithmetic: 2-bit only; two runs per wave: built for 4-bit codes, and for
// every width in the experiment build)
template <int C, int B>
void launch_encode_flat(bool direct, int eruns, dim3 grid, dim3 block, hipStream_t st,
                               const float *d_in, uint64_t nquad, uint8_t *o, bb_perm_t perm)
{
    if constexpr (B == 2) {
        if (direct) { hipLaunchKernelGGL((k_encode_flat<C, 2, true>), grid, block, 0, st, d_in, nquad, o, perm); return; }
    }
    if constexpr (B == 4 || BB_EXP) {
        if (eruns == 2) { hipLaunchKernelGGL((k_encode_flat<C, B, false, 2>), grid, block, 0, st, d_in, nquad, o, perm); return; }
    }
    hipLaunchKernelGGL((k_encode_flat<C, B, false>), grid, block, 0, st, d_in, nquad, o, perm);
}

// the byte table kernel with 16-byte loads staged through LDS: contiguous 2-bit
// output (the headline kernel; k_lds.h)
thread_local bool t_arena_probe = false;        // set around the launches of arena_probe (bb_arena.inc)

template <int BPS>
void launch_flat_lds(bool nt, dim3 grid, hipStream_t st, const bb_flat_args &a)
{
    with_nt(nt, [&](auto NT) {
        constexpr bool N = decltype(NT)::value;
        // 2-bit: whole 16-byte pieces go HBM -> LDS directly (global_load_lds_dwordx4,
        // GL = true): +1.7 % over load + ds_write_b128 at the headline size, bit-identical
        // (profiles/r04b_exp_glds.log); variant 19 of the experiment build = the register form
        if constexpr (BPS == 2) {
            if (t_arena_probe)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 2, 8, BB_LV_REG, 1, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
#if BB_EXP
            else if (g_tune_variant.load() == 19)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 2, 8, BB_LV_REG, 0, false>), grid, dim3(2 * BB_WAVE), 0, st, a);
            else if (g_tune_variant.load() == 21)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 4, 8, BB_LV_REG, 0, true>), grid, dim3(4 * BB_WAVE), 0, st, a);
            else if (g_tune_variant.load() == 22)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 1, 8, BB_LV_REG, 0, true>), grid, dim3(1 * BB_WAVE), 0, st, a);
            else if (g_tune_variant.load() == 23)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 2, 8, BB_LV_REG, 0, true, 1>), grid, dim3(2 * BB_WAVE), 0, st, a);
            else if (g_tune_variant.load() == 24)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 2, 8, BB_LV_REG, 0, true, 2>), grid, dim3(2 * BB_WAVE), 0, st, a);
            else if (g_tune_variant.load() == 25)
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 2, 8, BB_LV_REG, 0, true, 3>), grid, dim3(2 * BB_WAVE), 0, st, a);
#endif
            else
                hipLaunchKernelGGL((k_decode_flat_lds<2, N, 2, 8, BB_LV_REG, 0, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
        } else if constexpr (BPS == 4) {
            // 4-bit (round 4): direct-to-LDS too -- against k_decode_flat_lut +2.6-3.4 % on VDIF
            // frames and +6.7-8.5 % on GSB blocks at 16 GiB in, +1.3 % at 8 GiB, two boxes
            // (profiles/r04r_exp_glds5_box*.log); register staging lost 3-12 % in round 3
#if BB_EXP
            if (g_tune_variant.load() == 19)
                hipLaunchKernelGGL((k_decode_flat_lds<4, N, 2, 8>), grid, dim3(2 * BB_WAVE), 0, st, a);
            else
#endif
            hipLaunchKernelGGL((k_decode_flat_lds<4, N, 2, 8, BB_LV_REG, 0, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
        } else {
#if BB_EXP
            if (g_tune_variant.load() == 20)
                hipLaunchKernelGGL((k_decode_flat_lds<BPS, N, 2, 8, BB_LV_REG, 0, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
            else
#endif
            hipLaunchKernelGGL((k_decode_flat_lds<BPS, N, 2, 8>), grid, dim3(2 * BB_WAVE), 0, st, a);
        }
    });
}

#if BB_EXP
// the loader-wave kernel (k_burst.h): dynamic LDS up to the CU's 160 KiB
template <int NSTORE>
int launch_flat_burst(bool nt, dim3 grid, size_t lds, hipStream_t st, const bb_burst_args &a)
{
    int rc = BB_OK;
    with_nt(nt, [&](auto NT) {
        constexpr bool N = decltype(NT)::value;
        auto kern = k_decode_flat_burst<2, N, NSTORE>;
        static std::atomic<size_t> allowed{0};
        if (lds > allowed.load()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) { rc = hip_fail(hipGetLastError()); return; }
            allowed = lds;
        }
        hipLaunchKernelGGL(kern, grid, dim3((NSTORE + 1) * BB_WAVE), lds, st, a);
    });
    return rc;
}

#endif

// the same staging for contiguous 8-bit output: 2 waves x up to 16 tiles.  The
// product builds the int8 form with direct-to-LDS loads (3b below); the
// experiment build also the register-staged form and the table-level (VDIF
// 8-bit) instantiations it was measured against (profiles/r03zd_exp_flat8*.log,
// r04c_exp_glds2.log, r04d_exp_glds3_box*.log).
void launch_flat_lds8(int coder, bool nt, bool gl, dim3 grid, hipStream_t st, const bb_flat_args &a)
{
    with_nt(nt, [&](auto NT) {
        constexpr bool N = decltype(NT)::value;
#if BB_EXP
        if (coder == BB_CODER_INT && !gl) { hipLaunchKernelGGL((k_decode_flat_lds<8, N, 2, 16, BB_LV_INT8>), grid, dim3(2 * BB_WAVE), 0, st, a); return; }
        if (coder != BB_CODER_INT && !gl) { hipLaunchKernelGGL((k_decode_flat_lds<8, N, 2, 16, BB_LV_LDS>), grid, dim3(2 * BB_WAVE), 0, st, a); return; }
#endif
        (void)gl;
        if (coder != BB_CODER_INT) hipLaunchKernelGGL((k_decode_flat_lds<8, N, 2, 16, BB_LV_LDS, 0, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
        else hipLaunchKernelGGL((k_decode_flat_lds<8, N, 2, 16, BB_LV_INT8, 0, true>), grid, dim3(2 * BB_WAVE), 0, st, a);
    });
}

// the byte table kernel with dword loads handed out by ds_bpermute: contiguous
// 1- and 4-bit output (k_lut.h)
void launch_flat_lut(int bps, bool nt, dim3 grid, hipStream_t st, const bb_flat_args &a)
{
    with_nt(nt, [&](auto NT) {
        constexpr bool N = decltype(NT)::value;
        // (1-bit work items are at most 8 tiles per wave: the 8-tile instantiation
        // stays below 128 VGPRs, the 16-tile one needs 129 = one wave per SIMD less)
        // (the product sends 1-bit samples here; 2- and 4-bit ones go through k_decode_flat_lds,
        // their instantiations of this kernel stay in the experiment build for A/B: variant 16)
#if BB_EXP
        if (bps == 2) { hipLaunchKernelGGL((k_decode_flat_lut<2, N, 2, 16>), grid, dim3(2 * BB_WAVE), 0, st, a); return; }
        if (bps == 4) { hipLaunchKernelGGL((k_decode_flat_lut<4, N, 2, 16>), grid, dim3(2 * BB_WAVE), 0, st, a); return; }
#endif
        (void)bps;
        hipLaunchKernelGGL((k_decode_flat_lut<1, N, 2, 8>), grid, dim3(2 * BB_WAVE), 0, st, a);
    });
}

#if BB_EXP
#include "bb_exp.inc"
#endif

} // namespace

extern "C" {

int bb_abi_version(void) { return BB_ABI_VERSION; }

const char *bb_strerror(int code)
{
    switch (code) {
        case BB_OK:      return "success";
        case BB_EIO:     return "HIP runtime call failed";
        case BB_EINVAL:  return "invalid argument";
        case BB_ERANGE:  return "buffer too small or index out of range";
        case BB_ENOTSUP: return "unsupported coder / bits per sample / mode";
        default:         return "unknown error";
    }
}

int bb_last_hip_error(void) { return t_last_hip; }

const char *bb_last_kernel(void) { return t_last_kernel; }

int bb_init(void) { return ensure_init(); }

int bb_get_levels(int coder, int bps, float *h_out, size_t n)
{
    if (!h_out) return BB_EINVAL;
    if (!coder_supported(coder, bps)) return BB_ENOTSUP;
    if (n < ((size_t)1 << bps)) return BB_ERANGE;
    std::call_once(h_levels_once, fill_host_levels);
    memcpy(h_out, h_levels[coder][log2_bps(bps)], sizeof(float) << bps);
    return BB_OK;
}

int bb_get_encode_thresholds(float h_thr[3])
{
    if (!h_thr) return BB_EINVAL;
    std::call_once(h_levels_once, fill_host_levels);
    memcpy(h_thr, h_enc2_thr, sizeof(h_enc2_thr));
    return BB_OK;
}

int bb_tune(int knob, int value)
{
    switch (knob) {
        case BB_TUNE_BLOCKS:       g_tune_blocks = value;  return BB_OK;
        case BB_TUNE_TILE_ELEMS:   g_tune_tile_elems = value > 0 ? value : 8192; return BB_OK;
        case BB_TUNE_ENCODE_DIRECT: g_tune_encode_direct = value; return BB_OK;
        case BB_TUNE_TILES_PER_WAVE:
            g_tune_rows_tiles = (value >= 1 && value <= 8) ? value : 8;
#if BB_EXP
            g_tune_tpw = (value >= 1 && value <= 16) ? value : 12;
#endif
            return BB_OK;
        case BB_TUNE_GATHER_BYTES: g_tune_gather_bytes = value > 0 ? value : 8192; return BB_OK;
        case BB_TUNE_TILED_STAGE: g_tune_tiled_stage = value; return BB_OK;
        case BB_TUNE_SEG_TILES: g_tune_seg_tiles = (value >= 1 && value <= 4096) ? value : 0; return BB_OK;
        case BB_TUNE_GATHER_CHUNKS: g_tune_gather_chunks = value > 0 ? value : 32; return BB_OK;
        case BB_TUNE_MKBF_CHANNELS: g_tune_mkbf_tc = (value >= 2 && value <= 64 && !(value & 1)) ? value : 32; return BB_OK;
        case BB_TUNE_M4_WIDEN: g_tune_m4_widen = value; return BB_OK;
        case BB_TUNE_M4_TILES: g_tune_m4_tiles = (value >= 1 && value <= BB_M4_TPW) ? value : BB_M4_TPW; return BB_OK;
        case BB_TUNE_LUT_TILES: g_tune_lut_tpw = (value >= 1 && value <= 16) ? value : 0; return BB_OK;
        case BB_TUNE_SELECT_BYTES:
            if (value < 256 || value > 32768) return BB_EINVAL;
            g_tune_select_bytes = value; return BB_OK;
        case BB_TUNE_VDIF8_LDS_GIB: g_tune_vdif8_lds_gib = value < 0 ? 20 : value; return BB_OK;
        case BB_TUNE_TOUCH_MIB: g_tune_touch_mib = value < 0 ? touch_mib_default() : value; return BB_OK;
        case BB_TUNE_SELECT_PICK: g_tune_select_pick = value < 0 ? 1 : (value > 2 ? 2 : value); return BB_OK;
        case BB_TUNE_PICK_BYTES:
            if (value < 1024 || value > 32768) return BB_EINVAL;
            g_tune_pick_bytes = value; return BB_OK;
        case BB_TUNE_XPOSE: g_tune_xpose = value; return BB_OK;
        case BB_TUNE_XPOSE_ROWS: g_tune_xpose_rows = value == 64 ? 64 : value == 128 ? 128 : 0; return BB_OK;
        case BB_TUNE_ENCODE_RUNS: g_tune_encode_runs = (value == 1 || value == 2) ? value : 0; return BB_OK;
        case BB_TUNE_ENCODE_STRIPES: g_tune_encode_lw = (value >= 0 && value <= 10) ? value : 0; return BB_OK;
        case BB_TUNE_XPOSE_TC: g_tune_xpose_tc = (value == 64 || value == 32 || value == 16 || value == 8) ? value : 0; return BB_OK;
        case BB_TUNE_XPOSE_MIN_NC: g_tune_xpose_min_nc = value < 2 ? 2 : value; return BB_OK;
        case BB_TUNE_GATHER_GLDS: g_tune_gather_glds = value < 0 ? -1 : (value != 0); return BB_OK;
        case BB_TUNE_WORK_STRIPES: g_tune_order_lw = (value >= 0 && value <= 10) ? value : -1; return BB_OK;
#if BB_EXP
        case BB_TUNE_FLAT_VARIANT: g_tune_variant = value; return BB_OK;
        case BB_TUNE_COPY: g_tune_copy = value; return BB_OK;
        case BB_TUNE_M4_LDS: g_tune_m4_lds = value != 0; return BB_OK;
        case BB_TUNE_BURST: g_tune_burst = value; return BB_OK;
        case BB_TUNE_BURST_BYTES: g_tune_burst_bytes = (value >= 4096 && value <= 79360) ? (value & ~255) : 65536; return BB_OK;
        case BB_TUNE_BURST_PERIOD: g_tune_burst_period = value > 0 ? value : 0; return BB_OK;
        case BB_TUNE_BURST_WAVES: g_tune_burst_waves = (value == 3 || value == 7 || value == 15) ? value : 15; return BB_OK;
        case BB_TUNE_FLAT8_LDS: g_tune_flat8_lds = value; return BB_OK;
        case BB_TUNE_NT_STORES:    g_tune_nt = value;      return BB_OK;
        case BB_TUNE_NT_LOADS:     g_tune_nt_loads = value; return BB_OK;
        case BB_TUNE_TILES_PER_WAVE_8BIT: g_tune_tpw8 = (value >= 1 && value <= 32) ? value : 12; return BB_OK;
        case BB_TUNE_LDS_PAD: g_tune_lds_pad = (value > 0 && value <= 65536) ? value : 0; return BB_OK;
        case BB_TUNE_BYTE_LUT: g_tune_byte_lut = value; return BB_OK;
        case BB_TUNE_LUT_SMALL: g_tune_lut_small = value; return BB_OK;
        case BB_TUNE_OUT_STRIPE_W: g_tune_stripe_w = value > 0 ? value : 0; return BB_OK;
        case BB_TUNE_OUT_STRIPE_S: g_tune_stripe_s = value > 0 ? value : 0; return BB_OK;
        case BB_TUNE_FRONT_GROUP: g_tune_front_g = (value >= 1 && value <= (1 << 20)) ? value : 2048; return BB_OK;
        case BB_TUNE_FRONT_STEPS: g_tune_front_k = (value >= 1 && value <= (1 << 20)) ? value : 16; return BB_OK;
#endif
        default: return BB_EINVAL;
    }
}

#if BB_EXP
int bb_debug_trace(uint64_t *d_times)
{
    g_trace = d_times;
    return BB_OK;
}

// ---- host staging helpers (measurement only) -----------------------------------
// A file image that is mapped into the host's address space can be pinned where
// it lies and handed to the DMA engine, instead of being copied into a pinned
// buffer by host threads first (staging.py).
int bb_host_register(const void *h_ptr, size_t nbytes)
{
    if (!h_ptr || nbytes == 0) return BB_EINVAL;
    BB_HIP(hipHostRegister(const_cast<void *>(h_ptr), nbytes, hipHostRegisterDefault));
    return BB_OK;
}

int bb_host_unregister(const void *h_ptr)
{
    if (!h_ptr) return BB_EINVAL;
    BB_HIP(hipHostUnregister(const_cast<void *>(h_ptr)));
    return BB_OK;
}

int bb_copy_to_device(void *d_dst, const void *h_src, size_t nbytes, void *stream)
{
    if (nbytes == 0) return BB_OK;
    if (!d_dst || !h_src) return BB_EINVAL;
    BB_HIP(hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return BB_OK;
}
#endif

static int vdif_scan_impl(const void *d_buf, size_t nbytes, const bb_vdif_scan_params *p,
                          bb_frame_rec *d_recs, size_t nframes, int64_t *d_fill, size_t fill_n, void *stream)
{
    if (nframes == 0) {                                  // (an empty window: nothing to look at)
        if (d_fill && fill_n) BB_HIP(hipMemsetAsync(d_fill, 0xff, fill_n * sizeof(int64_t), (hipStream_t)stream));
        return BB_OK;
    }
    if (!d_buf || !p || !d_recs) return BB_EINVAL;
    if (p->header_nbytes != 32 && p->header_nbytes != 16) return BB_EINVAL;
    if (p->frame_nbytes < p->header_nbytes || (p->frame_nbytes & 7)) return BB_EINVAL;
    if ((p->first_offset & 3) || ((uintptr_t)d_buf & 3)) return BB_EINVAL;
    const uint64_t threads = (uint64_t)nframes * 8;
    const uint64_t blocks = (threads + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_vdif_scan, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)d_buf, (uint64_t)nbytes, *p, d_recs, (uint64_t)nframes, d_fill, (uint64_t)fill_n);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_vdif_scan(const void *d_buf, size_t nbytes, const bb_vdif_scan_params *p,
                 bb_frame_rec *d_recs, size_t nframes, void *stream)
{
    return vdif_scan_impl(d_buf, nbytes, p, d_recs, nframes, nullptr, 0, stream);
}

// records -> dense index (pre-filled by the caller's scan launch, or here) + verification, one launch
static int index_verify(const bb_frame_rec *d_recs, size_t nrecs, const int16_t *d_thread_slot, int nslot,
                        int64_t *d_src, size_t nframes_out, bool prefilled, uint32_t recs_per_index, size_t nstrict,
                        uint32_t *d_nbad, void *stream)
{
    if (!d_src || nslot < 1 || (nrecs && !d_recs) || recs_per_index == 0) return BB_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!prefilled && nframes_out)
        BB_HIP(hipMemsetAsync(d_src, 0xff, nframes_out * (size_t)nslot * sizeof(int64_t), st));
    if (nrecs == 0) return BB_OK;
    const uint64_t blocks = ((uint64_t)nrecs + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_index_verify, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, st, d_recs, (uint64_t)nrecs,
                       d_thread_slot, nslot, d_src, (uint64_t)nframes_out, recs_per_index, (uint64_t)nstrict, d_nbad);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_vdif_locate(const void *d_buf, size_t nbytes, const bb_vdif_scan_params *p,
                   int64_t *d_offsets, size_t cap, unsigned long long *d_count, void *stream)
{
    if (!d_buf || !p || !d_offsets || !d_count) return BB_EINVAL;
    if (p->header_nbytes != 32 && p->header_nbytes != 16) return BB_EINVAL;
    if (p->frame_nbytes < p->header_nbytes) return BB_EINVAL;
    if ((uintptr_t)d_buf & 15) return BB_EINVAL;           // 16-byte loads (bb_locate_sweep)
    if (nbytes < p->frame_nbytes || p->frame_nbytes < 32) return BB_OK;
    uint64_t blocks = (nbytes / 16 + BB_BLOCK * BB_LOCATE_U - 1) / (BB_BLOCK * BB_LOCATE_U);   // BB_LOCATE_U x 16 bytes per lane and iteration
    // (65536 workgroups: 5.1-5.8 TB/s on the 8 GiB image, 16384: 5.0-5.2, 131072: 4.3-4.4;
    // profiles/r04l_locate.log -- the workgroups' confirm phases overlap other workgroups' sweeps)
    const uint64_t lcap = g_tune_blocks.load() > 0 ? (uint64_t)g_tune_blocks.load() : BB_LOCATE_GRID;
    if (blocks > lcap) blocks = lcap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_vdif_locate, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)d_buf, (uint64_t)nbytes, *p, d_offsets, (uint64_t)cap, d_count);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_vdif_scan_at(const void *d_buf, size_t nbytes, const bb_vdif_scan_params *p,
                    const int64_t *d_offsets, size_t nframes, bb_frame_rec *d_recs, void *stream)
{
    if (!d_buf || !p || !d_offsets || !d_recs) return BB_EINVAL;
    if (p->header_nbytes != 32 && p->header_nbytes != 16) return BB_EINVAL;
    if ((uintptr_t)d_buf & 3) return BB_EINVAL;
    if (nframes == 0) return BB_OK;
    const uint64_t blocks = ((uint64_t)nframes + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_vdif_scan_at, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)d_buf, (uint64_t)nbytes, *p, d_offsets, d_recs, (uint64_t)nframes);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

static int mark5b_scan_impl(const void *d_buf, size_t nbytes, const bb_mark5b_scan_params *p,
                            const int64_t *d_offsets, bb_frame_rec *d_recs, size_t nframes,
                            void *stream)
{
    if (nframes == 0) return BB_OK;
    if (!d_buf || !p || !d_recs) return BB_EINVAL;
    if ((!d_offsets && (p->first_offset & 3)) || ((uintptr_t)d_buf & 3)) return BB_EINVAL;
    const uint64_t blocks = ((uint64_t)nframes + BB_WAVES_PER_BLOCK - 1) / BB_WAVES_PER_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_mark5b_scan, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)d_buf, (uint64_t)nbytes, *p, d_offsets, d_recs,
                       (uint64_t)nframes);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_mark5b_scan(const void *d_buf, size_t nbytes, const bb_mark5b_scan_params *p,
                   bb_frame_rec *d_recs, size_t nframes, void *stream)
{
    return mark5b_scan_impl(d_buf, nbytes, p, nullptr, d_recs, nframes, stream);
}

int bb_mark5b_scan_at(const void *d_buf, size_t nbytes, const bb_mark5b_scan_params *p,
                      const int64_t *d_offsets, size_t nframes, bb_frame_rec *d_recs, void *stream)
{
    if (!d_offsets && nframes) return BB_EINVAL;
    return mark5b_scan_impl(d_buf, nbytes, p, d_offsets, d_recs, nframes, stream);
}

int bb_mark5b_locate_stream(const void *d_buf, size_t nbytes, uint32_t w1_pattern, uint32_t w1_mask,
                            int64_t *d_offsets, size_t cap, unsigned long long *d_count, void *stream)
{
    if (!d_buf || !d_offsets || !d_count) return BB_EINVAL;
    if ((uintptr_t)d_buf & 15) return BB_EINVAL;           // 16-byte loads (bb_locate_sweep)
    if (nbytes < BB_M5B_FRAME) return BB_OK;
    uint64_t blocks = (nbytes / 16 + BB_BLOCK * BB_LOCATE_U - 1) / (BB_BLOCK * BB_LOCATE_U);   // BB_LOCATE_U x 16 bytes per lane and iteration
    if (blocks > (g_tune_blocks.load() > 0 ? (uint64_t)g_tune_blocks.load() : BB_LOCATE_GRID)) blocks = g_tune_blocks.load() > 0 ? (uint64_t)g_tune_blocks.load() : BB_LOCATE_GRID;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_mark5b_locate, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)d_buf, (uint64_t)nbytes, w1_pattern, w1_mask, d_offsets, (uint64_t)cap, d_count);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_mark5b_locate(const void *d_buf, size_t nbytes, int64_t *d_offsets, size_t cap,
                     unsigned long long *d_count, void *stream)
{
    return bb_mark5b_locate_stream(d_buf, nbytes, 0u, 0u, d_offsets, cap, d_count, stream);
}

int bb_verify_records(const bb_frame_rec *d_recs, size_t nrecs, int32_t first_index,
                      uint32_t recs_per_index, size_t nstrict, uint32_t *d_nbad, void *stream)
{
    if (!d_nbad || (nrecs && !d_recs) || recs_per_index == 0) return BB_EINVAL;
    if (nrecs == 0) return BB_OK;
    const uint64_t blocks = ((uint64_t)nrecs + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_verify_records, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       d_recs, (uint64_t)nrecs, first_index, recs_per_index, (uint64_t)nstrict, d_nbad);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_build_index(const bb_frame_rec *d_recs, size_t nrecs,
                   const int16_t *d_thread_slot, int nslot,
                   int64_t *d_src, size_t nframes_out, void *stream)
{
    if (!d_src || nslot < 1 || (nrecs && !d_recs)) return BB_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (nframes_out)
        BB_HIP(hipMemsetAsync(d_src, 0xff, nframes_out * (size_t)nslot * sizeof(int64_t), st));
    if (nrecs == 0) return BB_OK;
    const uint64_t blocks = ((uint64_t)nrecs + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_build_index, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, st,
                       d_recs, (uint64_t)nrecs, d_thread_slot, nslot, d_src, (uint64_t)nframes_out);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_decode_frames(const void *d_buf, size_t buf_nbytes,
                     const int64_t *d_src, size_t nframes,
                     const bb_decode_params *p,
                     float *d_out, size_t out_elems, void *stream)
{
    if (!p) return BB_EINVAL;
    if (!coder_supported(p->coder, p->bps)) return BB_ENOTSUP;
    if (nframes == 0) return BB_OK;
    if (!d_buf || !d_out) return BB_EINVAL;
    if (p->nslot < 1 || p->chunk < 1) return BB_EINVAL;
    if (p->payload_nbytes == 0 || (p->payload_nbytes & 3)) return BB_EINVAL;
    if (((uintptr_t)d_buf & 3) || ((uintptr_t)d_out & 15)) return BB_EINVAL;
    const uint64_t E = p->payload_nbytes * 8 / (uint64_t)p->bps;
    const uint64_t nfs = (uint64_t)nframes * (uint64_t)p->nslot;
    uint32_t lchunk = 0;
    int om = BB_OUT_FLAT;
    if (p->nslot > 1) {
        if (p->chunk & (p->chunk - 1)) return BB_ENOTSUP;   // VDIF nchan is 2^k
        while ((1u << lchunk) < (uint32_t)p->chunk) ++lchunk;
        if (E % (uint64_t)p->chunk) return BB_EINVAL;
        om = (p->chunk % 4 == 0) ? BB_OUT_ROWS4 : BB_OUT_SCATTER;
    }
    if (out_elems < nfs * E) return BB_ERANGE;
    if (!d_src) {
        if ((p->src0 & 3) || (p->src_stride & 3) || p->src0 < 0 || p->src_stride < 0) return BB_EINVAL;
        if (nfs && (uint64_t)p->src0 + (nfs - 1) * (uint64_t)p->src_stride + p->payload_nbytes > buf_nbytes)
            return BB_ERANGE;
    }
    if (nfs == 0) return BB_OK;
    int rc = ensure_init();
    if (rc) return rc;

    const int lb = log2_bps(p->bps);
    bb_flat_args a;
    a.buf = (const uint8_t *)d_buf;
    a.src = d_src;
    a.out = d_out;
    rc = device_levels(p->coder, lb, &a.tab);
    if (rc) return rc;
    a.nfs = nfs;
    a.ndw = p->payload_nbytes / 4;
    a.nseg = 1; a.seg_tiles = 0; a.tpw = 0;             // (set by the branch that launches)
    a.src0 = p->src0;
    a.src_stride = p->src_stride;
    a.nslot = (uint32_t)p->nslot;
    a.chunk = (uint32_t)p->chunk;
    a.lchunk = lchunk;
    a.fill_re = p->fill_re;
    a.fill_im = p->fill_im;
    a.complex_data = p->complex_data;
    // an index entry is followed only if its payload lies inside the buffer
    // (fixed-stride launches were checked above)
    a.src_lim = src_limit(buf_nbytes, p->payload_nbytes);
    a.perm = bb_perm_t{0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    const bool nt = tune_nt();
#if BB_EXP
    a.nt_loads = g_tune_nt_loads.load();
    a.trace = g_trace.load();
    a.stripe_w = 0;
    a.stripe_s = 0;
    if (g_tune_stripe_w.load() > 0 && om == BB_OUT_FLAT) {
        // experiment: the caller's buffer must hold (w - 1) * s + ceil(nfs / w) slots
        a.stripe_w = (uint32_t)g_tune_stripe_w.load();
        a.stripe_s = (uint64_t)g_tune_stripe_s.load();
        const uint64_t need = ((uint64_t)a.stripe_w - 1) * a.stripe_s + (nfs + a.stripe_w - 1) / a.stripe_w;
        if (out_elems < need * E) return BB_ERANGE;
    }
    {
        const int erc = expd::decode_frames(p, d_src, nframes, a, om, d_out, st);
        if (erc != 1) return erc;
    }
#endif

    const uint64_t ntiles = (a.ndw + 63) / 64;
    const uint64_t out_bytes = nfs * E * 4;
    const int tb = g_tune_blocks.load();

    // 1. Thread interleave through the LDS gather (k_gather.h): narrow chunks
    // (a thread's sample is less than 128 bytes of output -- through the rows
    // kernel, whose waves each write their own 16..64-byte pieces of every row,
    // chunks of 4 / 8 / 16 floats ran at 0.55 / 1.15 / 3.45 TB/s,
    // profiles/r01i_exp_interleave.log), any chunk with up to four thread
    // slots (one wave per slot leaves the rows kernel with 2-4 waves per
    // workgroup: 5.6 / 6.1 TB/s against 6.4 / 6.3,
    // profiles/r01i_exp_interleave_thr.log) and, since the staging rewrite of
    // round 2, 2-bit launches of 64 GiB of output and more whatever the chunk
    // (+1-3.5 % in every cell of 8-64 slots x 32-256 floats at 8 GiB of input,
    // profiles/r02at_exp_rows_vs_gather_2bit_8GiB.log; smaller launches and
    // other sample widths show no clear winner and stay with the rows kernel)
    const int gchunks = g_tune_gather_chunks.load();
    const bool big2 = p->bps == 2 && p->chunk <= 256 && out_bytes >= (64ull << 30);
    const bool gather_wide = om == BB_OUT_ROWS4
        && (gchunks == 32 ? (p->nslot <= 4 || p->chunk < 32 || big2) : p->chunk < gchunks);
    if ((om == BB_OUT_SCATTER || gather_wide) && d_src
        && (size_t)p->nslot * 528 + 1024 + 64 <= 48 * 1024) {
        bb_gather_args ga;
        ga.within = nullptr; ga.nsel = 0; ga.mag_row = ga.mag_sel = 0;
        ga.buf = a.buf; ga.src = d_src; ga.out = d_out; ga.tab = a.tab;
        ga.nframes = nframes; ga.ndw = a.ndw;
        ga.nslot = a.nslot; ga.chunk = a.chunk; ga.lchunk = a.lchunk;
        ga.src_lim = a.src_lim;
        // bytes staged per work item: 16 KiB, 4 KiB for 1-bit data whose items
        // expand 32-fold (profiles/r01i_exp_interleave_gb.log); knob value 8192 = this default
        size_t gbytes = (size_t)g_tune_gather_bytes.load();
        if (gbytes == 8192) gbytes = p->bps == 1 ? 4096 : 16384;
        uint32_t gt = (uint32_t)(gbytes / ((size_t)p->nslot * 256));
        if (gt < 1) gt = 1;
        if (gt > 32) gt = 32;
        if ((uint64_t)gt > ntiles) gt = (uint32_t)ntiles;
        ga.gtiles = gt;
        ga.ngroup = (uint32_t)((ntiles + gt - 1) / gt);
        ga.fill_re = a.fill_re; ga.fill_im = a.fill_im; ga.complex_data = a.complex_data;
        {
            const uint32_t rl = (uint32_t)p->nslot * (uint32_t)p->chunk;
            ga.lrow = -1;
            if ((rl & (rl - 1)) == 0) { ga.lrow = 0; while ((1u << ga.lrow) < rl) ++ga.lrow; }
        }
        ga.aligned = 1;
        ga.glds = g_tune_gather_glds.load() == 1;
        const size_t lds = ((size_t)p->nslot * (gt * 64 + 65) + 2 * p->nslot + 1) * 4 + 1024;
        // persistent grid: a workgroup walks about five work items (8 KiB of
        // payload each); one workgroup per item costs 15 %, a few thousand
        // long-running ones 5-10 % (profiles/r01f_exp_gather*.log)
        uint64_t gb = (uint64_t)nframes * ga.ngroup;
        ga.perm = make_perm(gb, out_bytes, BB_ORDER_GATHER);
        const uint64_t gcap = tb > 0 ? (uint64_t)tb : BB_GRID_CAP;
        if (gb > gcap) gb = gcap;
        if (gb > 0x7fffffffull) gb = 0x7fffffffull;
        const dim3 gg((unsigned)gb);
        launch_gather(p->bps, p->coder, nt, gg, lds, st, ga);
        BB_NOTE("k_decode_gather<%d,%s,%s,%s> grid %u gtiles %u lds %zu", p->bps, lv_name(p->bps, p->coder),
                nt ? "nt" : "plain", ga.lchunk >= 2 ? "wide" : "narrow", gg.x, ga.gtiles, lds);
        BB_HIP(hipGetLastError());
        return BB_OK;
    }

    if (om == BB_OUT_ROWS4) {
        // 2. thread interleave with wide chunks: one wave per thread slot, all
        // waves on the same (up to 8) tiles (k_decode_rows_pipe)
        const int nw = p->nslot >= 8 ? 8 : (p->nslot >= 4 ? 4 : 2);
        const uint64_t seg_max = (uint64_t)g_tune_rows_tiles.load();
        a.nseg = (ntiles + seg_max - 1) / seg_max;
        a.seg_tiles = (uint32_t)((ntiles + a.nseg - 1) / a.nseg);
        a.tpw = a.seg_tiles;
        const uint64_t sgroups = ((uint64_t)p->nslot + nw - 1) / nw;
        uint64_t b2 = (uint64_t)nframes * a.nseg * sgroups;
        a.perm = make_perm(b2, out_bytes, BB_ORDER_ROWS);
        const uint64_t cap = tb > 0 ? (uint64_t)tb : BB_GRID_CAP;
        if (b2 > cap) b2 = cap;
        const dim3 g2((unsigned)b2);
        launch_rows_pipe(p->bps, p->coder, nt, nw, g2, st, a);
        BB_NOTE("k_decode_rows_pipe<%d,%s,%s,%d,8,aligned> grid %u", p->bps, lv_name(p->bps, p->coder),
                nt ? "nt" : "plain", nw, g2.x);
        BB_HIP(hipGetLastError());
        return BB_OK;
    }

    if (om == BB_OUT_FLAT && p->bps <= 4) {
        // 3. contiguous 1-, 2- and 4-bit output: the byte table kernel (k_lut.h)
        // with SHORT work items, one per workgroup -- 32 KiB of OUTPUT each: 2
        // waves x 2 / 4 / 8 tiles for 1- / 2- / 4-bit samples, grid up to 2^23.
        // +2-4 % over 2 x 8-12 tiles on 131072 persistent workgroups at every
        // size from 2 to 8 GiB, for 8000-, 8192- and 10000-byte payloads
        // (profiles/r02ao_exp_tpw_grid*.log): many small workgroups in flight
        // overlap loads and stores better than a register pipeline, as for the
        // 8-bit kernels.  1-bit 6.58 -> 6.86 TB/s against 4 tiles, 4-bit 5.28 ->
        // 6.62 and +7 % over the plain kernel it used before
        // (profiles/r02ar_exp_1bit_items.log, r02ar_exp_4bit_lut.log)
        // 2-bit samples: k_decode_flat_lds (16-byte loads through LDS, at most 8 tiles
        // per wave); 1- and 4-bit: k_decode_flat_lut -- profiles/r03j_exp_lds.log
#if BB_EXP
        if (p->bps == 2 && g_tune_burst.load() != 0 && p->payload_nbytes >= 256) {
            // 3a. the loader-wave kernel (k_burst.h): items of up to one staging buffer
            bb_burst_args b;
            b.buf = a.buf; b.src = d_src; b.out = d_out; b.tab = a.tab;
            b.nfs = nfs; b.pbytes = p->payload_nbytes;
            b.buf_bytes = (uint32_t)g_tune_burst_bytes.load();
            const uint64_t fst = (b.pbytes + 30) & ~15ull;
            if (fst <= b.buf_bytes) {
                b.nseg = 1; b.seg_bytes = 0; b.fstride = (uint32_t)fst;
                uint64_t k = b.buf_bytes / fst;
                if (k > BB_BURST_MAXK) k = BB_BURST_MAXK;
                b.kpi = (uint32_t)k;
            } else {
                b.seg_bytes = (b.buf_bytes - 32) & ~255u;
                b.nseg = (b.pbytes + b.seg_bytes - 1) / b.seg_bytes;
                b.fstride = b.buf_bytes; b.kpi = 1;
            }
            b.magic = (uint32_t)((1ull << 32) / b.pbytes) + 1;
            const uint64_t nwork = nfs * b.nseg;
            b.nitems = (nwork + b.kpi - 1) / b.kpi;
            b.period = (uint32_t)g_tune_burst_period.load();
            b.src0 = a.src0; b.src_stride = a.src_stride;
            b.fill_re = a.fill_re; b.fill_im = a.fill_im; b.complex_data = a.complex_data;
            b.src_lim = a.src_lim;
            b.perm = make_perm(b.nitems, out_bytes);
            const int nst = g_tune_burst_waves.load();
            const size_t ldsb = BB_BURST_HEAD + 2 * (size_t)b.buf_bytes;
            uint64_t per_cu = (160 * 1024) / ldsb;
            const uint64_t by_waves = 32 / (uint64_t)(nst + 1);
            if (per_cu > by_waves) per_cu = by_waves;
            if (per_cu < 1) per_cu = 1;
            uint64_t gb = tb > 0 ? (uint64_t)tb : 256 * per_cu;
            if (gb > b.nitems) gb = b.nitems;
            const dim3 gg((unsigned)gb);
            int brc;
            if (nst == 3) brc = launch_flat_burst<3>(nt, gg, ldsb, st, b);
            else if (nst == 7) brc = launch_flat_burst<7>(nt, gg, ldsb, st, b);
            else brc = launch_flat_burst<15>(nt, gg, ldsb, st, b);
            if (brc) return brc;
            BB_NOTE("k_decode_flat_burst<2,%s,%d> grid %u pieces/item %u buffer %u period %u", nt ? "nt" : "plain", nst,
                    gg.x, b.kpi, b.buf_bytes, b.period);
            BB_HIP(hipGetLastError());
            return BB_OK;
        }
#endif
        bool lds = p->bps == 2 || p->bps == 4;
#if BB_EXP
        if (g_tune_variant.load() == 15 || (g_tune_variant.load() >= 19 && g_tune_variant.load() <= 25)) lds = true;        // A/B: force either kernel for every sample width
        if (g_tune_variant.load() == 16) lds = false;
#endif
        // (2-bit through k_decode_flat_lds: 6 tiles per wave -- 0.851-0.859 of the peak with
        // and without an index on three boxes, 4 tiles: 0.850-0.858 with, 0.837-0.848
        // without; profiles/r04d_exp_glds3_box*.log.  Knob 0 = these defaults.)
        const int lt_knob = g_tune_lut_tpw.load();
        // (4-bit through k_decode_flat_lds: 4 tiles per wave; r04r_exp_glds5_box*.log)
        int lut_tiles = lt_knob == 0 ? ((lds && p->bps == 2) ? 6 : (lds && p->bps == 4) ? 4 : 4 * p->bps / 2) : lt_knob * p->bps / 2;
        lut_tiles = lut_tiles < 1 ? 1 : lut_tiles > (lds ? 8 : 16) ? (lds ? 8 : 16) : lut_tiles;
        uint64_t nwv = 2;                                   // waves per workgroup
#if BB_EXP
        if (lds && p->bps == 2 && g_tune_variant.load() == 21) nwv = 4;     // A/B: 4 / 1 waves per workgroup
        if (lds && p->bps == 2 && g_tune_variant.load() == 22) nwv = 1;
#endif
        const uint64_t seg_max = nwv * (uint64_t)lut_tiles;
        a.nseg = (ntiles + seg_max - 1) / seg_max;
        a.seg_tiles = (uint32_t)((ntiles + a.nseg - 1) / a.nseg);
        a.tpw = (uint32_t)((a.seg_tiles + nwv - 1) / nwv);
        uint64_t b2 = nfs * a.nseg;
        a.perm = make_perm(b2, out_bytes, BB_ORDER_FLAT);
        const uint64_t cap = tb > 0 ? (uint64_t)tb : (1ull << 23);
        if (b2 > cap) b2 = cap;
        const dim3 g2((unsigned)b2);
        if (lds) {
#if BB_EXP
            if (p->bps == 1) launch_flat_lds<1>(nt, g2, st, a);
            else
#endif
            if (p->bps == 4) launch_flat_lds<4>(nt, g2, st, a);
            else launch_flat_lds<2>(nt, g2, st, a);
            const char *gl = "";
#if BB_EXP
            if (p->bps != 1 && g_tune_variant.load() == 19) gl = ",regs";
            if (p->bps == 1 && g_tune_variant.load() == 20) gl = ",glds";
#endif
            BB_NOTE("k_decode_flat_lds<%d,%s,2,8%s> grid %u tiles/wave %u", p->bps, nt ? "nt" : "plain", gl, g2.x, a.tpw);
        } else {
            launch_flat_lut(p->bps, nt, g2, st, a);
            BB_NOTE("k_decode_flat_lut<%d,%s,2,16> grid %u tiles/wave %u", p->bps, nt ? "nt" : "plain", g2.x, a.tpw);
        }
        BB_HIP(hipGetLastError());
        return BB_OK;
    }

    // 3b. contiguous int8 output (DADA, GSB, GUPPI real; 20 % of the traffic is
    // reads): k_decode_flat_lds<8> with direct-to-LDS loads, 2 waves x 4 tiles = 8
    // KiB of output per work item.  Against the plain kernel (4): +4.7 % at 8 GiB in
    // on three boxes, +1.2 % at 31 GiB (profiles/r04d_exp_glds3_box*.log,
    // r04c_exp_glds2.log); with 8 / 16 tiles per wave or register staging it
    // loses.  VDIF 8-bit frames (table levels) stay with the plain kernel up to
    // 20 GiB of payload: -6 % at 8 GiB, +3-4 % at 31 GiB (the switch below).
    {
        bool lds8 = om == BB_OUT_FLAT && p->bps == 8 && p->coder == BB_CODER_INT;
        bool gl8 = true;
        int t8 = 4;
        // VDIF 8-bit frames (table levels), round 5: BY SIZE.  k_decode_flat_lds<8,LDS,glds> with 16
        // tiles per wave against the plain kernel, 8032-byte frames (profiles/r05g_exp_vdif8_size.log):
        // 0.5-12 GiB in: -0.1 .. -3.4 %; 16 GiB +0.4 %, 24 GiB +2.5 %, 31 GiB +4.1 % (0.791 -> 0.823 of
        // the peak).  From 20 GiB of payload on it takes the staged kernel.
        if (om == BB_OUT_FLAT && p->bps == 8 && p->coder != BB_CODER_INT
            && nfs * (uint64_t)p->payload_nbytes >= ((uint64_t)g_tune_vdif8_lds_gib.load() << 30)) {
            lds8 = true;
            t8 = 16;
        }
#if BB_EXP
        const int f8 = g_tune_flat8_lds.load();                 // 1: staged for every coder, knobs apply; 2: plain kernel
        if (f8 == 1) {
            lds8 = om == BB_OUT_FLAT && p->bps == 8;
            t8 = (g_tune_lut_tpw.load() ? g_tune_lut_tpw.load() : 4) * 4;
            t8 = t8 < 1 ? 1 : t8 > 16 ? 16 : t8;
            gl8 = g_tune_variant.load() == 20;
        } else if (f8 == 2) lds8 = false;
#endif
        if (lds8) {
            const uint64_t seg_max = 2ull * (uint64_t)t8;
            a.nseg = (ntiles + seg_max - 1) / seg_max;
            a.seg_tiles = (uint32_t)((ntiles + a.nseg - 1) / a.nseg);
            a.tpw = (a.seg_tiles + 1) / 2;
            uint64_t b2 = nfs * a.nseg;
            a.perm = make_perm(b2, out_bytes, BB_ORDER_FLAT);
            const uint64_t cap = tb > 0 ? (uint64_t)tb : (1ull << 23);
            if (b2 > cap) b2 = cap;
            const dim3 g2((unsigned)b2);
            launch_flat_lds8(p->coder, nt, gl8, g2, st, a);
            BB_NOTE("k_decode_flat_lds<8,%s,%s,2,16%s> grid %u tiles/wave %u", lv_name(p->bps, p->coder), nt ? "nt" : "plain",
                    gl8 ? ",glds" : "", g2.x, a.tpw);
            BB_HIP(hipGetLastError());
            return BB_OK;
        }
    }

    // 4. the plain kernel (k_decode_flat): one workgroup of four waves per work
    // item, loads and stores in the same iteration, uncapped grid.  8-bit
    // samples with contiguous output read 20 % of their traffic instead of 6 %,
    // and many small workgroups overlap that better than a register pipeline
    // (profiles/r01i_exp_int8_v0.log: 5.34-5.37 -> 5.55-5.67 TB/s int8,
    // 5.10-5.37 -> 5.44-5.58 VDIF 8-bit); 16 tiles per workgroup (4 KiB in, 16
    // KiB out) instead of 32: +7 % at 8 GiB, +0-3.5 % at 31 GiB
    // (profiles/r02ba_exp_int8_seg.log).  Also the general fallback: thread
    // interleave without an index, or with more slots than the gather stages.
    {
        const int seg_knob = g_tune_seg_tiles.load();
        const uint64_t seg_plain = seg_knob ? (uint64_t)seg_knob : (p->bps == 8 ? 16u : (uint64_t)BB_SEG_TILES);
        a.nseg = (ntiles + seg_plain - 1) / seg_plain;
        // split a frame-slot's tiles evenly over its work items and a work item's
        // tiles evenly over the four waves (a 10000-byte Mark 5B payload is 40
        // tiles: 2 items x 20 tiles x 5 per wave, not 32 + 8)
        a.seg_tiles = (uint32_t)((ntiles + a.nseg - 1) / a.nseg);
        a.tpw = (a.seg_tiles + BB_WAVES_PER_BLOCK - 1) / BB_WAVES_PER_BLOCK;
        const uint64_t nwork = nfs * a.nseg;
        a.perm = make_perm(nwork, out_bytes, BB_ORDER_FLAT);
        uint64_t blocks = nwork;
        if (tb > 0 && blocks > (uint64_t)tb) blocks = (uint64_t)tb;
        if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
        const dim3 grid((unsigned)blocks);
        launch_flat(p->bps, p->coder, om, nt, grid, st, a);
        BB_NOTE("k_decode_flat<%d,%s,%d,%s> grid %u", p->bps, lv_name(p->bps, p->coder), om,
                nt ? "nt" : "plain", grid.x);
        BB_HIP(hipGetLastError());
        return BB_OK;
    }
}

// Argument checks and work geometry of bb_decode_frames_select, shared with
// bb_decode_frames_select_check (no device is touched here).
struct select_geom { uint32_t lchunk, gt; uint64_t R, ntiles; size_t lds; };

static int select_geometry(const bb_decode_params *p, int nwithin, select_geom *g)
{
    if (!p) return BB_EINVAL;
    if (!coder_supported(p->coder, p->bps)) return BB_ENOTSUP;
    if (p->nslot < 1 || p->chunk < 1 || nwithin < 1 || nwithin > 4096) return BB_EINVAL;
    if (p->payload_nbytes == 0 || (p->payload_nbytes & 3)) return BB_EINVAL;
    if (p->chunk & (p->chunk - 1)) return BB_ENOTSUP;
    uint32_t lchunk = 0;
    while ((1u << lchunk) < (uint32_t)p->chunk) ++lchunk;
    const uint64_t E = p->payload_nbytes * 8 / (uint64_t)p->bps;
    if (E % (uint64_t)p->chunk) return BB_EINVAL;
    const uint64_t ntiles = (p->payload_nbytes / 4 + 63) / 64;
    // stage at most 16 KiB of payload per work item (all slots together) and
    // at most 16 tiles of a slot -- a single slot is fastest with 4 KiB per
    // item, eight with 16 KiB (profiles/r02ac_exp_select.log) -- in groups of
    // equal size (a 10000-byte Mark 5B payload: 14 + 14 + 12 tiles, not 32 + 8)
    uint32_t gt = (uint32_t)((size_t)g_tune_select_bytes.load() / ((size_t)p->nslot * 256));
    if (gt < 1) gt = 1;
    if (gt > 16) gt = 16;
    if ((uint64_t)gt > ntiles) gt = (uint32_t)ntiles;
    gt = (uint32_t)((ntiles + (ntiles + gt - 1) / gt - 1) / ((ntiles + gt - 1) / gt));
    // a group's elements must be whole rows: gt * 2048 / bps elements, chunk a power of two
    while (gt > 1 && ((uint64_t)gt * (2048 / p->bps)) % (uint64_t)p->chunk) --gt;
    if (((uint64_t)gt * (2048 / p->bps)) % (uint64_t)p->chunk) return BB_ENOTSUP;
    const size_t lds = ((size_t)p->nslot * (gt * 64 + 65) + 2 * p->nslot + 1) * 4 + 1024 + (size_t)nwithin * 4;
    if (lds > 64 * 1024) return BB_ENOTSUP;
    g->lchunk = lchunk; g->gt = gt; g->R = E >> lchunk; g->ntiles = ntiles; g->lds = lds;
    return BB_OK;
}

int bb_touch(const void *d_buf, size_t nbytes, void *stream)
{
    if (!d_buf) return nbytes ? BB_EINVAL : BB_OK;
    // whole 16-byte chunks between the first and the last aligned address inside the range
    const uintptr_t lo = ((uintptr_t)d_buf + 15) & ~(uintptr_t)15, hi = ((uintptr_t)d_buf + nbytes) & ~(uintptr_t)15;
    if (hi <= lo) return BB_OK;
    const uint64_t nchunk = (uint64_t)(hi - lo) / 16;
    const uint64_t blocks = (nchunk + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    hipLaunchKernelGGL(k_touch, dim3((unsigned)blocks), dim3(BB_BLOCK), 0, (hipStream_t)stream,
                       (const bb_u4 *)lo, nchunk, (uint32_t *)nullptr);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

// A read window that fits the memory-side cache is read through once on the decode's stream
// while the scan and the index run on theirs (or in front of them, without a side stream): the
// decode then takes its input from that cache and runs nearer the rate of its stores.  fh.read()
// of 2^12 / 2^13 / 2^14 / 2^15 frames of 8 KiB: +4.3-6.1 / +4.6-7.5 / +5.1-7.4 / +5.1-5.2 % on two
// boxes, within +-1 % at 2^16 (too large: not read) and below 16 MiB (profiles/r06cx_, r06cy_exp_touch_read.log;
// with NONTEMPORAL loads in the pre-read the reads got 2.4-3.8 % slower: those go past the
// cache, r06cw_).
static int touch_window(const void *d_buf, size_t nbytes, void *stream)
{
    const uint64_t lim = (uint64_t)g_tune_touch_mib.load() << 20;
    if (lim == 0 || nbytes < (16u << 20) || nbytes > lim) return BB_OK;     // (below 16 MiB the launch costs what it saves)
    return bb_touch(d_buf, nbytes, stream);
}

int bb_vdif_read_window(const void *d_buf, size_t nbytes,
                        const bb_vdif_scan_params *scan, size_t nframes,
                        const int16_t *d_thread_slot, size_t nsets,
                        const bb_decode_params *dec,
                        const int32_t *d_within, int nwithin,
                        bb_frame_rec *d_recs, int64_t *d_src,
                        float *d_out, size_t out_elems,
                        uint32_t recs_per_index, size_t nstrict, uint32_t *d_nbad,
                        void *verified, void *scan_stream, void *stream)
{
    if (!scan || !dec || !d_src || dec->nslot < 1) return BB_EINVAL;
    if (scan_stream && !verified) return BB_EINVAL;        // (the decode's stream waits for that event)
    void *ss = scan_stream ? scan_stream : stream;
    { const int trc = touch_window(d_buf, nbytes, stream); if (trc != BB_OK) return trc; }
    // three launches: scan (which also pre-sets the index to -1), index + verification, decode
    bb_vdif_scan_params sp = *scan;
    sp.set_nframes = (int32_t)recs_per_index;               // frame sets in file order (bbdecode.h)
    int rc = vdif_scan_impl(d_buf, nbytes, &sp, d_recs, nframes, d_src, nsets * (size_t)dec->nslot, ss);
    if (rc != BB_OK) return rc;
    rc = index_verify(d_recs, nframes, d_thread_slot, dec->nslot, d_src, nsets, true, recs_per_index ? recs_per_index : 1,
                      nstrict, d_nbad, ss);
    if (rc != BB_OK) return rc;
    if (verified) BB_HIP(hipEventRecord((hipEvent_t)verified, (hipStream_t)ss));
    if (scan_stream) BB_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)verified, 0));
    if (nwithin > 0)
        return bb_decode_frames_select(d_buf, nbytes, d_src, nsets, dec, d_within, nwithin, d_out, out_elems, stream);
    return bb_decode_frames(d_buf, nbytes, d_src, nsets, dec, d_out, out_elems, stream);
}

int bb_mark5b_read_window(const void *d_buf, size_t nbytes,
                          const bb_mark5b_scan_params *scan, size_t nframes, size_t n,
                          const bb_decode_params *dec,
                          const int32_t *d_within, int nwithin,
                          bb_frame_rec *d_recs, int64_t *d_src,
                          float *d_out, size_t out_elems,
                          size_t nstrict, uint32_t *d_nbad, void *verified, void *scan_stream, void *stream)
{
    if (!scan || !dec) return BB_EINVAL;
    if (scan_stream && !verified) return BB_EINVAL;
    void *ss = scan_stream ? scan_stream : stream;
    { const int trc = touch_window(d_buf, nbytes, stream); if (trc != BB_OK) return trc; }
    int rc = bb_mark5b_scan(d_buf, nbytes, scan, d_recs, nframes, ss);
    if (rc != BB_OK) return rc;
    rc = index_verify(d_recs, nframes, nullptr, 1, d_src, n, false, 1, nstrict, d_nbad, ss);
    if (rc != BB_OK) return rc;
    if (verified) BB_HIP(hipEventRecord((hipEvent_t)verified, (hipStream_t)ss));
    if (scan_stream) BB_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)verified, 0));
    if (nwithin > 0)
        return bb_decode_frames_select(d_buf, nbytes, d_src, n, dec, d_within, nwithin, d_out, out_elems, stream);
    return bb_decode_frames(d_buf, nbytes, d_src, n, dec, d_out, out_elems, stream);
}

int bb_mark4_read_window(const void *d_buf, size_t nbytes,
                         const bb_mark4_scan_params *scan, size_t nframes, size_t n,
                         const bb_mark4_decode_params *dec, int nout,
                         bb_frame_rec *d_recs, int64_t *d_src,
                         float *d_out, size_t out_elems,
                         size_t nstrict, uint32_t *d_nbad, void *verified, void *scan_stream, void *stream)
{
    if (!scan || !dec) return BB_EINVAL;
    if (scan_stream && !verified) return BB_EINVAL;
    void *ss = scan_stream ? scan_stream : stream;
    { const int trc = touch_window(d_buf, nbytes, stream); if (trc != BB_OK) return trc; }
    int rc = bb_mark4_scan(d_buf, nbytes, scan, d_recs, nframes, ss);
    if (rc != BB_OK) return rc;
    rc = index_verify(d_recs, nframes, nullptr, 1, d_src, n, false, 1, nstrict, d_nbad, ss);
    if (rc != BB_OK) return rc;
    if (verified) BB_HIP(hipEventRecord((hipEvent_t)verified, (hipStream_t)ss));
    if (scan_stream) BB_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)verified, 0));
    if (nout > 0)
        return bb_decode_mark4_select(d_buf, nbytes, d_src, n, dec, nout, d_out, out_elems, stream);
    return bb_decode_mark4(d_buf, nbytes, d_src, n, dec, d_out, out_elems, stream);
}

int bb_fetch_counter(const uint32_t *d_counter, uint32_t *h_value, void *after, void *side_stream)
{
    if (!d_counter || !h_value) return BB_EINVAL;
    hipStream_t st = (hipStream_t)side_stream;
    if (after) BB_HIP(hipStreamWaitEvent(st, (hipEvent_t)after, 0));
    BB_HIP(hipMemcpyAsync(h_value, d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    BB_HIP(hipStreamSynchronize(st));
    return BB_OK;
}

// float32 samples: a strided frame copy (k_copy.h).  EXTENSION -- the reference
// has no NBIT 32 decoder (dada/payload.py:40-41).
int bb_copy_frames(const void *d_buf, size_t buf_nbytes, size_t nframes, uint64_t nbytes_per_frame,
                   int64_t src0, int64_t src_stride, void *d_out, size_t out_nbytes, void *stream)
{
    if (nframes == 0 || nbytes_per_frame == 0) { BB_NOTE("none"); return BB_OK; }
    if (!d_buf || !d_out) return BB_EINVAL;
    if ((nbytes_per_frame & 3) || (src0 & 3) || (src_stride & 3) || src0 < 0 || src_stride < 0
        || (reinterpret_cast<uintptr_t>(d_buf) & 3) || (reinterpret_cast<uintptr_t>(d_out) & 3)) return BB_EINVAL;
    if (nframes > 1 && (uint64_t)src_stride < nbytes_per_frame && src_stride != 0) return BB_EINVAL;
    if ((uint64_t)nframes > (~0ull) / nbytes_per_frame || (uint64_t)nframes * nbytes_per_frame > out_nbytes) return BB_ERANGE;
    if ((uint64_t)src0 + (uint64_t)(nframes - 1) * (uint64_t)src_stride + nbytes_per_frame > buf_nbytes) return BB_ERANGE;
    int rc = ensure_init();
    if (rc) return rc;
    bb_copy_args a;
    a.buf = (const uint8_t *)d_buf; a.out = (uint8_t *)d_out;
    a.nframes = nframes; a.n = nbytes_per_frame;
    int nl = 4, ntl = 1;
#if BB_EXP
    { const int cv = g_tune_copy.load(); if (cv) { nl = cv & 0xff; ntl = (cv >> 8) & 1; } }
    if (nl != 2 && nl != 4 && nl != 8 && nl != 16) nl = 4;
#endif
    const uint64_t item = (uint64_t)nl * BB_BLOCK * 16;
    a.nseg = (nbytes_per_frame + item - 1) / item;
    a.src0 = src0; a.src_stride = src_stride;
    const uint64_t nwork = (uint64_t)nframes * a.nseg;
    a.perm = make_perm(nwork, (uint64_t)nframes * nbytes_per_frame);
    const bool v16 = !((reinterpret_cast<uintptr_t>(d_buf) | reinterpret_cast<uintptr_t>(d_out) | (uint64_t)src0
                        | (uint64_t)src_stride | nbytes_per_frame) & 15);
    const int tb = g_tune_blocks.load();
    uint64_t blocks = nwork;
    const uint64_t cap = tb > 0 ? (uint64_t)tb : (1ull << 23);
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = tune_nt();
    with_nt(nt, [&](auto NT) {
        constexpr bool N = decltype(NT)::value;
        if (!v16) { hipLaunchKernelGGL((k_copy_frames<N, false>), grid, dim3(BB_BLOCK), 0, st, a); return; }
#if BB_EXP
        if (nl == 2 && ntl)  { hipLaunchKernelGGL((k_copy_frames<N, true, 2, true>), grid, dim3(BB_BLOCK), 0, st, a); return; }
        if (nl == 8 && ntl)  { hipLaunchKernelGGL((k_copy_frames<N, true, 8, true>), grid, dim3(BB_BLOCK), 0, st, a); return; }
        if (nl == 16 && ntl) { hipLaunchKernelGGL((k_copy_frames<N, true, 16, true>), grid, dim3(BB_BLOCK), 0, st, a); return; }
        if (nl == 4 && !ntl) { hipLaunchKernelGGL((k_copy_frames<N, true, 4, false>), grid, dim3(BB_BLOCK), 0, st, a); return; }
        if (nl == 8 && !ntl) { hipLaunchKernelGGL((k_copy_frames<N, true, 8, false>), grid, dim3(BB_BLOCK), 0, st, a); return; }
#endif
        hipLaunchKernelGGL((k_copy_frames<N, true>), grid, dim3(BB_BLOCK), 0, st, a);
    });
    (void)ntl;
    BB_NOTE("k_copy_frames<%s,%s,%d,%s> grid %u", nt ? "nt" : "plain", v16 ? "16B" : "4B", nl, ntl ? "ntload" : "load", grid.x);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_decode_frames_select_check(const bb_decode_params *p, int nwithin)
{
    select_geom g;
    return select_geometry(p, nwithin, &g);
}

int bb_decode_frames_select(const void *d_buf, size_t buf_nbytes,
                            const int64_t *d_src, size_t nframes,
                            const bb_decode_params *p,
                            const int32_t *d_within, int nwithin,
                            float *d_out, size_t out_elems, void *stream)
{
    select_geom g;
    int rc = select_geometry(p, nwithin, &g);
    if (rc) return rc;
    if (nframes == 0) return BB_OK;
    if (!d_buf || !d_out || !d_src || !d_within) return BB_EINVAL;
    if (((uintptr_t)d_buf & 3) || ((uintptr_t)d_out & 3)) return BB_EINVAL;
    const uint32_t lchunk = g.lchunk, gt = g.gt;
    const uint64_t R = g.R, ntiles = g.ntiles;
    if (out_elems < (uint64_t)nframes * R * (uint64_t)p->nslot * (uint64_t)nwithin) return BB_ERANGE;
    rc = ensure_init();
    if (rc) return rc;
    const int lb = log2_bps(p->bps);
    bb_gather_args ga;
    ga.buf = (const uint8_t *)d_buf; ga.src = d_src; ga.out = d_out;
    rc = device_levels(p->coder, lb, &ga.tab);
    if (rc) return rc;
    ga.nframes = nframes; ga.ndw = p->payload_nbytes / 4;
    ga.nslot = (uint32_t)p->nslot; ga.chunk = (uint32_t)p->chunk; ga.lchunk = lchunk;
    ga.src_lim = src_limit(buf_nbytes, p->payload_nbytes);
    ga.gtiles = gt;
    ga.ngroup = (uint32_t)((ntiles + gt - 1) / gt);
    ga.fill_re = p->fill_re; ga.fill_im = p->fill_im; ga.complex_data = p->complex_data;
    ga.lrow = -1;
    ga.aligned = 1;
    ga.glds = g_tune_gather_glds.load() != 0;
    ga.within = d_within; ga.nsel = (uint32_t)nwithin;
    {
        // floats a work item writes at most, and the two divisors of its index walk
        const uint64_t qmax = ((uint64_t)gt * (2048 / p->bps) >> lchunk) * (uint64_t)p->nslot * (uint64_t)nwithin;
        const uint64_t drow = (uint64_t)p->nslot * (uint64_t)nwithin, dsel = (uint64_t)nwithin;
        ga.mag_row = (drow > 1 && qmax * drow < (1ull << 32)) ? (uint32_t)((1ull << 32) / drow + 1) : 0;
        ga.mag_sel = (dsel > 1 && qmax * dsel < (1ull << 32)) ? (uint32_t)((1ull << 32) / dsel + 1) : 0;
    }
    {
        // k_decode_pick (k_pick.h): whole-byte thread samples, an output row of 4..256
        // floats that is a power of two, at most 32 slots, whole rows per staged piece
        const uint64_t rowlen = (uint64_t)p->nslot * (uint64_t)nwithin;
        const uint32_t rowbits = (uint32_t)p->bps << lchunk;
        // Measured at the bench's shape (8 threads x 16 complex channels, 8 GiB in; profiles/r05f_exp_pick.log):
        // 1 of 16 channels 0.517 -> 0.769 of the peak on bytes moved, 2 of 16 0.688 -> 0.769 (4 KiB staged per
        // item; 8 KiB 0.739, 16 KiB 0.719), 4 of 16 0.737 -> 0.734 and 8 of 16 0.756 -> 0.747 at best: the wave
        // items win where the stores are few, so selections of up to an eighth of a thread sample take them
        // (BB_TUNE_SELECT_PICK = 2: whenever the conditions hold).
        const int pick_mode = g_tune_select_pick.load();
        if (pick_mode && (pick_mode == 2 || (uint64_t)nwithin * 8 <= (uint64_t)p->chunk)
            && !(rowbits & 7) && rowlen >= 4 && rowlen <= 256 && !(rowlen & (rowlen - 1))
            && p->nslot <= 32 && !((uintptr_t)d_out & 15)) {
            const uint32_t rowbytes = rowbits >> 3;
            uint32_t sb = ((uint32_t)g_tune_pick_bytes.load() / (uint32_t)p->nslot) & ~255u;
            if (sb > 4096) sb = 4096;
            if ((uint64_t)sb > ((p->payload_nbytes + 255) & ~255ull)) sb = (uint32_t)((p->payload_nbytes + 255) & ~255ull);
            if (sb >= 256 && rowbytes <= 256 && sb % rowbytes == 0 && p->payload_nbytes % rowbytes == 0
                && ((R * rowlen) % 4) == 0) {
                bb_pick_args pa;
                pa.buf = ga.buf; pa.src = d_src; pa.out = d_out; pa.tab = ga.tab; pa.within = d_within;
                pa.nframes = nframes; pa.pbytes = p->payload_nbytes; pa.src_lim = ga.src_lim;
                pa.nslot = (uint32_t)p->nslot; pa.nsel = (uint32_t)nwithin;
                pa.lrowlen = 0;
                while ((1ull << pa.lrowlen) < rowlen) ++pa.lrowlen;
                pa.rowbytes = rowbytes; pa.sb = sb;
                pa.nitem = (uint32_t)((p->payload_nbytes + sb - 1) / sb);
                pa.pitch = sb + 80;                          // 16 bytes of misalignment + a bank skew between the slots' rows
                pa.fill_re = p->fill_re; pa.fill_im = p->fill_im; pa.complex_data = p->complex_data;
                const uint64_t nwork = (uint64_t)nframes * pa.nitem;
                pa.perm = make_perm(nwork, (uint64_t)nframes * R * rowlen * 4);
                uint64_t gbp = (nwork + BB_PICK_NW - 1) / BB_PICK_NW;
                const int tbp = g_tune_blocks.load();
                const uint64_t capp = tbp > 0 ? (uint64_t)tbp : (BB_GRID_CAP << 3);
                if (gbp > capp) gbp = capp;
                const size_t ldsp = (size_t)BB_PICK_NW * pa.nslot * pa.pitch;
                launch_pick(p->bps, p->coder, tune_nt(), dim3((unsigned)gbp), ldsp, (hipStream_t)stream, pa);
                BB_NOTE("k_decode_pick<%d,%s,%s,%d> grid %u items of %u B x %u slots, select %d of %d", p->bps,
                        lv_name(p->bps, p->coder), tune_nt() ? "nt" : "plain", BB_PICK_NW, (unsigned)gbp, sb, pa.nslot,
                        nwithin, p->chunk);
                BB_HIP(hipGetLastError());
                return BB_OK;
            }
        }
    }
    const size_t lds = g.lds;
    uint64_t gb = (uint64_t)nframes * ga.ngroup;
    ga.perm = make_perm(gb, (uint64_t)nframes * R * p->nslot * nwithin * 4);
    const int tb = g_tune_blocks.load();
    const uint64_t gcap = tb > 0 ? (uint64_t)tb : BB_GRID_CAP;
    if (gb > gcap) gb = gcap;
    const dim3 gg((unsigned)gb);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = tune_nt();
    // float4 stores when every work item's output starts on a 16-byte boundary
    // and is a multiple of four floats: whole frame sets and whole groups are
    const uint64_t row_floats = (uint64_t)p->nslot * (uint64_t)nwithin;
    const uint64_t group_rows = ((uint64_t)gt * (2048 / p->bps)) >> lchunk;
    const bool v4 = ((R * row_floats) % 4 == 0) && ((group_rows * row_floats) % 4 == 0)
                    && !((uintptr_t)d_out & 15);
    launch_gather_select(p->bps, p->coder, nt, v4, gg, lds, st, ga);
    BB_NOTE("k_decode_gather_select<%d,%s,%s,%s> grid %u gtiles %u select %d of %d", p->bps, lv_name(p->bps, p->coder),
            nt ? "nt" : "plain", v4 ? "float4" : "scalar", gg.x, gt, nwithin, p->chunk);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

static int mark4_scan_impl(const void *d_buf, size_t nbytes, const bb_mark4_scan_params *p,
                           const int64_t *d_offsets, bb_frame_rec *d_recs, size_t nframes,
                           void *stream)
{
    if (nframes == 0) return BB_OK;
    if (!d_buf || !p || !d_recs) return BB_EINVAL;
    if (p->ntrack != 16 && p->ntrack != 32 && p->ntrack != 64) return BB_ENOTSUP;
    if ((!d_offsets && (p->first_offset & (p->ntrack / 8 - 1))) || ((uintptr_t)d_buf & 7)) return BB_EINVAL;
    if (p->frame_qms < 0) return BB_EINVAL;
    if (nframes == 0) return BB_OK;
    const uint64_t blocks = ((uint64_t)nframes + BB_WAVES_PER_BLOCK - 1) / BB_WAVES_PER_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    const dim3 grid((unsigned)blocks), block(BB_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *b = (const uint8_t *)d_buf;
    switch (p->ntrack) {
        case 16: hipLaunchKernelGGL(k_mark4_scan<16>, grid, block, 0, st, b, (uint64_t)nbytes, *p, d_offsets, d_recs, (uint64_t)nframes); break;
        case 32: hipLaunchKernelGGL(k_mark4_scan<32>, grid, block, 0, st, b, (uint64_t)nbytes, *p, d_offsets, d_recs, (uint64_t)nframes); break;
        default: hipLaunchKernelGGL(k_mark4_scan<64>, grid, block, 0, st, b, (uint64_t)nbytes, *p, d_offsets, d_recs, (uint64_t)nframes); break;
    }
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_mark4_scan(const void *d_buf, size_t nbytes, const bb_mark4_scan_params *p,
                  bb_frame_rec *d_recs, size_t nframes, void *stream)
{
    return mark4_scan_impl(d_buf, nbytes, p, nullptr, d_recs, nframes, stream);
}

int bb_mark4_scan_at(const void *d_buf, size_t nbytes, const bb_mark4_scan_params *p,
                     const int64_t *d_offsets, size_t nframes, bb_frame_rec *d_recs, void *stream)
{
    if (!d_offsets && nframes) return BB_EINVAL;
    return mark4_scan_impl(d_buf, nbytes, p, d_offsets, d_recs, nframes, stream);
}

int bb_mark4_locate(const void *d_buf, size_t nbytes, int ntrack, int64_t *d_offsets, size_t cap,
                    unsigned long long *d_count, void *stream)
{
    if (!d_buf || !d_offsets || !d_count) return BB_EINVAL;
    if (ntrack != 16 && ntrack != 32 && ntrack != 64) return BB_ENOTSUP;
    if ((uintptr_t)d_buf & 15) return BB_EINVAL;           // 16-byte loads (bb_locate_sweep)
    if (nbytes < (size_t)ntrack * 2500) return BB_OK;
    uint64_t blocks = (nbytes / 16 + BB_BLOCK * BB_LOCATE_U - 1) / (BB_BLOCK * BB_LOCATE_U);   // BB_LOCATE_U x 16 bytes per lane and iteration
    if (blocks > (g_tune_blocks.load() > 0 ? (uint64_t)g_tune_blocks.load() : BB_LOCATE_GRID)) blocks = g_tune_blocks.load() > 0 ? (uint64_t)g_tune_blocks.load() : BB_LOCATE_GRID;
    if (blocks == 0) blocks = 1;
    const dim3 grid((unsigned)blocks), block(BB_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *b = (const uint8_t *)d_buf;
    switch (ntrack) {
        case 16: hipLaunchKernelGGL(k_mark4_locate<16>, grid, block, 0, st, b, (uint64_t)nbytes, d_offsets, (uint64_t)cap, d_count); break;
        case 32: hipLaunchKernelGGL(k_mark4_locate<32>, grid, block, 0, st, b, (uint64_t)nbytes, d_offsets, (uint64_t)cap, d_count); break;
        default: hipLaunchKernelGGL(k_mark4_locate<64>, grid, block, 0, st, b, (uint64_t)nbytes, d_offsets, (uint64_t)cap, d_count); break;
    }
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_mark4_header_crc(const void *d_buf, size_t nbytes, int ntrack, const int64_t *d_offsets,
                        int64_t first_offset, size_t nframes, uint64_t *d_bad_tracks, void *stream)
{
    if (nframes == 0) return BB_OK;
    if (!d_buf || !d_bad_tracks) return BB_EINVAL;
    if (ntrack != 16 && ntrack != 32 && ntrack != 64) return BB_ENOTSUP;
    if (!d_offsets && first_offset < 0) return BB_EINVAL;
    const uint64_t blocks = ((uint64_t)nframes + BB_BLOCK - 1) / BB_BLOCK;
    if (blocks > 0x7fffffffull) return BB_ERANGE;
    const dim3 grid((unsigned)blocks), block(BB_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *b = (const uint8_t *)d_buf;
    switch (ntrack) {
        case 16: hipLaunchKernelGGL(k_mark4_header_crc<16>, grid, block, 0, st, b, (uint64_t)nbytes, d_offsets, first_offset, (uint64_t)nframes, d_bad_tracks); break;
        case 32: hipLaunchKernelGGL(k_mark4_header_crc<32>, grid, block, 0, st, b, (uint64_t)nbytes, d_offsets, first_offset, (uint64_t)nframes, d_bad_tracks); break;
        default: hipLaunchKernelGGL(k_mark4_header_crc<64>, grid, block, 0, st, b, (uint64_t)nbytes, d_offsets, first_offset, (uint64_t)nframes, d_bad_tracks); break;
    }
    BB_HIP(hipGetLastError());
    return BB_OK;
}

// Narrow streams as 64-bit super-words.  A 16- or 32-track stream word is 2 or
// 4 bytes: a wave's coalesced load of one word per lane moves 128 or 256
// bytes, and a store pass covers as few outputs per word.  64 / ntrack
// consecutive words ARE one little-endian 64-bit word whose output j (of 32)
// is output j % (ntrack/2) of narrow word j / (ntrack/2) -- bit position
// + ntrack * that word's number -- so whenever a unit's word count and fill
// prefix are multiples of 64 / ntrack (frames: 20000 and 160; payloads: 19840)
// the 64-track kernel decodes it with widened maps, 8 bytes per lane.
static bool m4_widen(const bb_mark4_decode_params *p, int nout, bb_mark4_decode_params *q, int *nout_q)
{
    const int r = 64 / p->ntrack;
    if (r == 1 || g_tune_m4_widen.load() == 0 || (p->nwords % (uint64_t)r) || (p->fill_words % (uint64_t)r)
            || nout * r > 32)
        return false;
    *q = *p;
    q->ntrack = 64;
    q->nwords = p->nwords / (uint64_t)r;
    q->fill_words = p->fill_words / (uint64_t)r;
    for (int w = 0; w < r; ++w)
        for (int k = 0; k < nout; ++k) {
            q->sign_bit[w * nout + k] = (uint8_t)(p->sign_bit[k] + w * p->ntrack);
            q->mag_bit[w * nout + k] = (uint8_t)(p->mag_bit[k] + w * p->ntrack);
        }
    *nout_q = nout * r;
    return true;
}

// shared by bb_decode_mark4 (nout = ntrack / 2, pipelined kernel) and
// bb_decode_mark4_select (any nout, LDS-staged kernel); arguments are checked
static int m4_decode(const void *d_buf, size_t buf_nbytes, const int64_t *d_src, size_t nframes,
                     const bb_mark4_decode_params *p_in, int nout_in, bool select,
                     float *d_out, void *stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    bb_mark4_decode_params wide;
    const bb_mark4_decode_params *p = p_in;
    int nout = nout_in;
    if (m4_widen(p_in, nout_in, &wide, &nout)) p = &wide;
    const uint64_t E = p->nwords * (uint64_t)nout;
    bb_m4_args a;
    a.buf = (const uint8_t *)d_buf;
    a.src = d_src;
    a.out = d_out;
    a.nframes = nframes;
    a.nwords = p->nwords;
    a.fill_words = p->fill_words;
    const uint64_t ntiles = (p->nwords + 63) / 64;
    const uint64_t m4_seg = (uint64_t)BB_WAVES_PER_BLOCK * (uint64_t)g_tune_m4_tiles.load();
    a.nseg = (ntiles + m4_seg - 1) / m4_seg;
    a.seg_tiles = (uint32_t)((ntiles + a.nseg - 1) / a.nseg);
    a.tpw = (a.seg_tiles + BB_WAVES_PER_BLOCK - 1) / BB_WAVES_PER_BLOCK;
    a.src0 = p->src0;
    a.src_stride = p->src_stride;
    memset(a.sign_bit, 0, sizeof(a.sign_bit));
    memset(a.mag_bit, 0, sizeof(a.mag_bit));
    memcpy(a.sign_bit, p->sign_bit, (size_t)nout);
    memcpy(a.mag_bit, p->mag_bit, (size_t)nout);
    a.fill = p->fill;
    a.hi = h_levels[BB_CODER_VDIF][1][3];
    // (the unit's size in bytes is the same before and after widening)
    a.src_lim = src_limit(buf_nbytes, p_in->nwords * ((uint64_t)p_in->ntrack / 8));
    uint64_t blocks = (uint64_t)nframes * a.nseg;
    a.perm = make_perm(blocks, (uint64_t)nframes * E * 4);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = tune_nt();
    const dim3 block(BB_BLOCK);
    if (!select) {
        const int tb = g_tune_blocks.load();
        // (one work item per workgroup up to 2^23: +1 % over 131072 persistent
        // workgroups at 8 GiB, equal at 2 GiB -- profiles/r02ao_exp_short_items.log)
        const uint64_t cap = tb > 0 ? (uint64_t)tb : (1ull << 23);
        if (blocks > cap) blocks = cap;
        const dim3 grid((unsigned)blocks);
#define BB_M4(N) with_nt(nt, [&](auto NT) { \
            hipLaunchKernelGGL((k_decode_mark4<N, decltype(NT)::value>), grid, block, 0, st, a); })
        bool m4lds = false;
#if BB_EXP
        // experiment (BB_TUNE_M4_LDS 1): 64-bit words staged in LDS by direct-to-LDS loads -- slower
        // than the shuffle form at 8 GiB (0.81 vs 0.83), +2.7 % at 2 GiB: profiles/r04s_exp_m4lds.log
        m4lds = p->ntrack == 64 && g_tune_m4_lds.load() != 0;
        if (m4lds) {
            with_nt(nt, [&](auto NT) { hipLaunchKernelGGL((k_decode_mark4_lds<decltype(NT)::value>), grid, block, 0, st, a); });
        } else
#endif
        switch (p->ntrack) {
            case 16: BB_M4(16); break;
            case 32: BB_M4(32); break;
            default: BB_M4(64); break;
        }
#undef BB_M4
        BB_NOTE("k_decode_mark4%s<%d,%s> grid %u%s", m4lds ? "_lds" : "", p->ntrack, nt ? "nt" : "plain", grid.x,
                p == &wide ? " (narrow words as 64-bit super-words)" : "");
    } else {
        if (blocks > (1ull << 30)) blocks = 1ull << 30;     // no pipeline to fill: one work item per workgroup
        const dim3 grid((unsigned)blocks);
        // float4 stores need every unit to start on a 16-byte boundary
        const bool v4 = (E % 4 == 0) && (((uintptr_t)d_out & 15) == 0);
#define BB_M4S(N) with_nt(nt, [&](auto NT) { \
        if (v4) hipLaunchKernelGGL((k_decode_mark4_select<N, decltype(NT)::value, true>), grid, block, 0, st, a, (uint32_t)nout); \
        else    hipLaunchKernelGGL((k_decode_mark4_select<N, decltype(NT)::value, false>), grid, block, 0, st, a, (uint32_t)nout); })
        switch (p->ntrack) {
            case 16: BB_M4S(16); break;
            case 32: BB_M4S(32); break;
            default: BB_M4S(64); break;
        }
#undef BB_M4S
        BB_NOTE("k_decode_mark4_select<%d,%s,%s> nout %d grid %u%s", p->ntrack, nt ? "nt" : "plain",
                v4 ? "v4" : "scalar", nout, grid.x,
                p == &wide ? " (narrow words as 64-bit super-words)" : "");
    }
    BB_HIP(hipGetLastError());
    return BB_OK;
}

static int m4_check(const void *d_buf, size_t buf_nbytes, const int64_t *d_src, size_t nframes,
                    const bb_mark4_decode_params *p, int nout, const float *d_out, size_t out_elems,
                    unsigned out_align)
{
    if (!d_buf || !d_out) return BB_EINVAL;
    if (p->nwords == 0 || p->fill_words > p->nwords) return BB_EINVAL;
    if (((uintptr_t)d_buf & 7) || ((uintptr_t)d_out & (out_align - 1))) return BB_EINVAL;
    const uint64_t wbytes = (uint64_t)p->ntrack / 8;
    for (int j = 0; j < nout; ++j)
        if (p->sign_bit[j] >= p->ntrack || p->mag_bit[j] >= p->ntrack) return BB_EINVAL;
    if (out_elems < (uint64_t)nframes * p->nwords * (uint64_t)nout) return BB_ERANGE;
    if (!d_src) {
        if (p->src0 < 0 || p->src_stride < 0 || (p->src0 % (int64_t)wbytes) || (p->src_stride % (int64_t)wbytes))
            return BB_EINVAL;
        if ((uint64_t)p->src0 + ((uint64_t)nframes - 1) * (uint64_t)p->src_stride + p->nwords * wbytes > buf_nbytes)
            return BB_ERANGE;
    }
    return BB_OK;
}

int bb_decode_mark4(const void *d_buf, size_t buf_nbytes,
                    const int64_t *d_src, size_t nframes,
                    const bb_mark4_decode_params *p,
                    float *d_out, size_t out_elems, void *stream)
{
    if (!p) return BB_EINVAL;
    if (p->ntrack != 16 && p->ntrack != 32 && p->ntrack != 64) return BB_ENOTSUP;
    if (nframes == 0) return BB_OK;
    const int rc = m4_check(d_buf, buf_nbytes, d_src, nframes, p, p->ntrack / 2, d_out, out_elems, 16);
    if (rc) return rc;
    return m4_decode(d_buf, buf_nbytes, d_src, nframes, p, p->ntrack / 2, false, d_out, stream);
}

int bb_decode_mark4_select(const void *d_buf, size_t buf_nbytes,
                           const int64_t *d_src, size_t nframes,
                           const bb_mark4_decode_params *p, int nout,
                           float *d_out, size_t out_elems, void *stream)
{
    if (!p) return BB_EINVAL;
    if (p->ntrack != 16 && p->ntrack != 32 && p->ntrack != 64) return BB_ENOTSUP;
    if (nout < 1 || nout > 32) return BB_EINVAL;
    if (nframes == 0) return BB_OK;
    const int rc = m4_check(d_buf, buf_nbytes, d_src, nframes, p, nout, d_out, out_elems, 4);
    if (rc) return rc;
    return m4_decode(d_buf, buf_nbytes, d_src, nframes, p, nout, true, d_out, stream);
}

int bb_decode_i8_tiled(const void *d_buf, size_t buf_nbytes,
                       const int64_t *d_src, size_t nframes,
                       const bb_tiled_params *p,
                       float *d_out, size_t out_elems, void *stream)
{
    if (!p) return BB_EINVAL;
    if (p->layout < 0 || p->layout > 2) return BB_ENOTSUP;
    if (p->npol < 1 || p->nchan < 1 || p->t_hi > p->ntime || p->t_lo > p->t_hi) return BB_EINVAL;
    if (p->layout == BB_LAYOUT_MKBF && (p->ntime % 256)) return BB_EINVAL;
    const uint64_t rows = p->t_hi - p->t_lo;
    if (nframes == 0 || rows == 0) return BB_OK;
    if (!d_buf || !d_out) return BB_EINVAL;
    if (((uintptr_t)d_buf & 1) || ((uintptr_t)d_out & 15)) return BB_EINVAL;
    const uint64_t rowlen = (uint64_t)p->npol * p->nchan * 2;           // floats per output time
    if (out_elems < (uint64_t)nframes * rows * rowlen) return BB_ERANGE;
    // a channel RANGE of the stored channels: the payload offsets point at the
    // first kept channel, strides follow the stored count
    if (p->nchan_stored < 0 || (p->nchan_stored && p->nchan_stored < p->nchan && !p->d_chan_map)) return BB_EINVAL;
    const uint64_t ncs = p->nchan_stored ? (uint64_t)p->nchan_stored : (uint64_t)p->nchan;
    // a selection (channel list and / or one of two polarisations): offsets
    // point at the payload start, strides follow the stored counts
    if (p->npol_stored < 0 || p->pol_first < 0) return BB_EINVAL;
    const uint64_t nps = p->npol_stored ? (uint64_t)p->npol_stored : (uint64_t)p->npol;
    if ((uint64_t)p->pol_first + (uint64_t)p->npol > nps) return BB_EINVAL;
    const bool selecting = p->d_chan_map != nullptr || nps != (uint64_t)p->npol;
    if (p->d_chan_map && !p->nchan_stored) return BB_EINVAL;
    // bytes from the first kept channel of the first time to the end of the last kept one
    uint64_t payload = p->ntime * nps * ncs * 2;
    if (!selecting && ncs != (uint64_t)p->nchan) {
        const uint64_t cut = ncs - (uint64_t)p->nchan;              // channels not entered
        payload -= (p->layout == BB_LAYOUT_GUPPI_CF ? cut * p->ntime * (uint64_t)p->npol
                    : p->layout == BB_LAYOUT_MKBF ? cut * 256 : cut * (uint64_t)p->npol) * 2;
    }
    if (!d_src) {
        if (p->src0 < 0 || p->src_stride < 0 || (p->src0 & 1) || (p->src_stride & 1)) return BB_EINVAL;
        if ((uint64_t)p->src0 + ((uint64_t)nframes - 1) * (uint64_t)p->src_stride + payload > buf_nbytes)
            return BB_ERANGE;
    }
    bb_tiled_args a;
    a.buf = (const uint8_t *)d_buf;
    a.src = d_src;
    a.out = d_out;
    a.nframes = nframes;
    a.t_lo = p->t_lo;
    a.t_hi = p->t_hi;
    a.src0 = p->src0;
    a.src_stride = p->src_stride;
    a.npol = (uint32_t)p->npol;
    a.nchan = (uint32_t)p->nchan;
    a.fill_re = p->fill_re;
    a.fill_im = p->fill_im;
    a.src_lim = src_limit(buf_nbytes, payload);
    a.nps = (uint32_t)nps;
    a.pf = (uint32_t)p->pol_first;
    a.cmap = p->d_chan_map;
    const uint64_t T = p->ntime, np_ = nps, nc = (uint64_t)p->nchan;
    switch (p->layout) {
        case BB_LAYOUT_GUPPI_CF: a.tb = T ? T : 1; a.sh = 0; a.st = np_; a.sp = 1; a.sc = T * np_; break;
        case BB_LAYOUT_MKBF:     a.tb = 256; a.sh = np_ * ncs * 256; a.st = 1; a.sp = ncs * 256; a.sc = 256; break;
        default:                 a.tb = T ? T : 1; a.sh = 0; a.st = ncs * np_; a.sp = 1; a.sc = np_; break;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool nt = tune_nt();
    const int tb = g_tune_blocks.load();
    // Fast form (k_xpose.h) for the common geometry: every input run 16-byte
    // aligned, at least 8 channels (a selection: 2).  Fixed stride only (offsets from an index
    // cannot be checked for alignment here).
    {
        const uint64_t npd = (uint64_t)p->npol;                          // polarisations decoded
        const uint64_t rows = (p->t_hi - p->t_lo) * npd;                 // output rows per frame
        // (a selection may keep as few as two channels; without one, blocks of
        // fewer than 8 stay with k_tiled.h)
        bool ok = g_tune_xpose.load() != 0 && !d_src && (selecting ? nc >= 2 : nc >= (uint64_t)g_tune_xpose_min_nc.load()) && !(nc & 1)
                  && !((uintptr_t)d_buf & 15) && !(p->src0 & 15) && !(p->src_stride & 15)
                  && (npd == np_ || (np_ == 2 && npd == 1));
        if (p->layout == BB_LAYOUT_GUPPI_CF)
            ok = ok && ((T * np_ * 2) % 16 == 0) && ((p->t_lo * np_ * 2) % 16 == 0);
        else if (p->layout == BB_LAYOUT_MKBF)
            ok = ok && (np_ == 1 || np_ == 2) && (p->t_lo % 8 == 0);
        else
            ok = ok && np_ == 2 && (p->d_chan_map ? true : (nc % 4 == 0)) && (ncs % 4 == 0);
        if (!ok && selecting) return BB_ENOTSUP;
        // a channel LIST of time-first blocks: whole rows by direct-to-LDS loads, the
        // kept channels picked out of LDS (k_tfpick.h; round 3: dword gathers in k_xpose.h)
        if (ok && p->layout == BB_LAYOUT_GUPPI_TF && p->d_chan_map && g_tune_xpose_tc.load() == 0
            && ncs * 4 <= BB_TFPICK_STAGE && nc <= BB_TFPICK_MAXSEL) {
            bb_tfpick_args b;
            b.buf = a.buf; b.out = d_out; b.cmap = p->d_chan_map;
            b.nframes = nframes; b.t_lo = p->t_lo; b.t_hi = p->t_hi;
            b.src0 = p->src0; b.src_stride = p->src_stride; b.src_lim = a.src_lim;
            b.rb = (uint32_t)(ncs * 4); b.nsel = (uint32_t)nc; b.npd = (uint32_t)npd; b.pf = (uint32_t)p->pol_first;
            b.tt = BB_TFPICK_STAGE / b.rb;
            const uint64_t ntt = ((p->t_hi - p->t_lo) + b.tt - 1) / b.tt;
            if (ntt > 0xffffffffull) return BB_ERANGE;
            b.ntt = (uint32_t)ntt;
            const uint32_t halfp = b.nsel / 2, ppt = halfp * b.npd;
            b.magic_ppt = (uint32_t)((1ull << 32) / ppt) + 1;
            b.magic_half = (uint32_t)((1ull << 32) / halfp) + 1;
            b.fill_re = p->fill_re; b.fill_im = p->fill_im;
            uint64_t blocks = (uint64_t)nframes * ntt;
            b.perm = make_perm(blocks, (uint64_t)nframes * (p->t_hi - p->t_lo) * rowlen * 4);
            const uint64_t cap = tb > 0 ? (uint64_t)tb : 0x7fffffffull;
            if (blocks > cap) blocks = cap;
            const dim3 grid((unsigned)blocks), block(BB_BLOCK);
            with_nt(nt, [&](auto NT) {
                constexpr bool N = decltype(NT)::value;
                hipLaunchKernelGGL((k_decode_i8_tf_pick<N>), grid, block, 0, st, b);
            });
            BB_NOTE("k_decode_i8_tf_pick<%s> grid %u tiles of %u times, %u of %u channels", nt ? "nt" : "plain",
                    grid.x, b.tt, b.nsel, (unsigned)ncs);
            BB_HIP(hipGetLastError());
            return BB_OK;
        }
        if (ok) {
            // Tile shape (the tile keeps its 16 KiB of input; a narrower tile is
            // longer along the input's contiguous axis).  Blocks of 64 channels, 8 and
            // 31 GiB of input, all eight shapes in one process on two boxes
            // (profiles/r03zb_exp_xpose_tc*.log):
            //   channels first  32 channels x 256 elements (512-byte input runs): 6.48 /
            //                   6.65 TB/s against 6.21 / 6.55 with 64 x 128 (16 x 512 the same)
            //   time first      16 channels x 256 times: 6.55 / 6.68 against 6.44 / 6.55
            //   MKBF heaps      64 channels x 64 rows: 6.47 / 6.63 against 6.20 / 6.35 with 128
            // and never wider than the channels there are (8 at least: rows of 64 bytes).
            const int xr = g_tune_xpose_rows.load(), xt = g_tune_xpose_tc.load();
            const uint64_t fit = nc > 32 ? 64u : nc > 16 ? 32u : nc > 8 ? 16u : 8u;
            // (time-first blocks with a channel map load single dwords: wide tiles
            // reuse the cache lines of a time's row, 8.6 against 10.1 ms for 64 mapped channels)
            const uint64_t pref = p->layout == BB_LAYOUT_GUPPI_CF ? 32u
                                : (p->layout == BB_LAYOUT_MKBF || p->d_chan_map) ? 64u : 16u;
            const uint64_t xtc = xt ? (uint64_t)xt : (fit < pref ? fit : pref);
            const uint64_t xrows = xr ? (uint64_t)xr : (p->layout == BB_LAYOUT_MKBF && xtc == 64) ? 64u : 128u;
            const uint64_t rt = xrows * 64 / xtc;
            const uint64_t rpt = p->layout == BB_LAYOUT_MKBF ? rt : rt / (np_ / npd);
            const uint64_t ntt = (rows + rpt - 1) / rpt, nct = (nc + xtc - 1) / xtc;
            if (ntt > 0xffffffffull) return BB_ERANGE;
            a.ntt = (uint32_t)ntt; a.nct = (uint32_t)nct;
            a.tt = (uint32_t)(rt / npd); a.tc = (uint32_t)xtc; a.tcp = 2 * ((uint32_t)xtc + 1);
            uint64_t blocks = (uint64_t)nframes * ntt * nct;
            a.perm = make_perm(blocks, (uint64_t)nframes * (p->t_hi - p->t_lo) * rowlen * 4, BB_ORDER_TILED);
            // one tile per workgroup: with 20 % of the traffic being reads the
            // dispatcher overlaps loads and stores of many small workgroups
            // better than a persistent pipelined grid does (16 GiB of input,
            // profiles/r02g_exp_i8_grid.log: 5.16-5.31 TB/s with 131072
            // workgroups, 5.62-5.66 with one per tile; the flat int8 kernel
            // behaves the same way)
            const uint64_t cap = tb > 0 ? (uint64_t)tb : 0x7fffffffull;
            if (blocks > cap) blocks = cap;
            const dim3 grid((unsigned)blocks), block(BB_BLOCK);
#define BB_XP1(L, R) \
                if (xtc == 8)        hipLaunchKernelGGL((k_decode_i8_xpose<L, N, R, 8>), grid, block, 0, st, a); \
                else if (xtc == 16)  hipLaunchKernelGGL((k_decode_i8_xpose<L, N, R, 16>), grid, block, 0, st, a); \
                else if (xtc == 32)  hipLaunchKernelGGL((k_decode_i8_xpose<L, N, R, 32>), grid, block, 0, st, a); \
                else                 hipLaunchKernelGGL((k_decode_i8_xpose<L, N, R, 64>), grid, block, 0, st, a);
#define BB_XP(L) with_nt(nt, [&](auto NT) { \
                constexpr bool N = decltype(NT)::value; \
                if (xrows == 64) { BB_XP1(L, 64) } else { BB_XP1(L, 128) } })
            switch (p->layout) {
                case BB_LAYOUT_GUPPI_CF: BB_XP(0); break;
                case BB_LAYOUT_MKBF:     BB_XP(1); break;
                default:                 BB_XP(2); break;
            }
#undef BB_XP1
#undef BB_XP
            BB_NOTE("k_decode_i8_xpose<%d,%s,%d,%d> grid %u tiles %u x %u per frame", p->layout, nt ? "nt" : "plain",
                    (int)xrows, (int)xtc, grid.x, a.ntt, a.nct);
            BB_HIP(hipGetLastError());
            return BB_OK;
        }
    }
    // tile: up to 64 channels (even count so float4 pieces pair up), and as
    // many times as keep the tile near 8192 elements (16 KiB in, 64 KiB out)
    uint32_t tc = (uint32_t)(nc < 64 ? nc : 64);
    if (tc > 1 && (tc & 1)) tc += 1;
    // MKBF and GUPPI time-first stage the tile in input order and permute on
    // the LDS read side (k_decode_i8_stage); BB_TUNE_TILED_STAGE 0 = old kernel
    const bool stage = p->layout != BB_LAYOUT_GUPPI_CF && g_tune_tiled_stage.load() != 0;
    uint32_t tile_elems = (uint32_t)g_tune_tile_elems.load();
    // MKBF: 32 channels x 128 times keeps the input runs at 256 bytes and the
    // tile at 16 KiB of LDS (more workgroups per CU overlap load and store phases)
    if (stage && p->layout == BB_LAYOUT_MKBF && tc > (uint32_t)g_tune_mkbf_tc.load())
        tc = (uint32_t)g_tune_mkbf_tc.load();
    uint32_t tt = tile_elems / (uint32_t)(np_ * tc);
    if (tt < 1) tt = 1;
    if (tt > 1024) tt = 1024;
    if (p->layout == BB_LAYOUT_MKBF) { if (tt > 256) tt = 256; while (256 % tt) --tt; }
    if ((uint64_t)tt > rows) tt = (uint32_t)rows;
    a.tt = tt;
    a.tc = tc;
    size_t lds;
    if (stage) {
        // LDS row = one input run: tt times (MKBF) or tc * npol elements (time-first)
        const uint32_t run = p->layout == BB_LAYOUT_MKBF ? tt : tc * (uint32_t)np_;
        uint32_t pitch_dw = (run + 1) / 2;
        if ((pitch_dw & 1) == 0) pitch_dw += 1;         // odd number of dwords
        a.tcp = 2 * pitch_dw;
        lds = (size_t)(p->layout == BB_LAYOUT_MKBF ? np_ * tc : tt) * a.tcp * sizeof(uint16_t);
    } else {
        uint32_t pitch_dw = ((tc + 1) / 2) + 1;         // dwords per LDS row
        if ((pitch_dw & 1) == 0) pitch_dw += 1;         // odd: strided writes hit distinct banks
        a.tcp = 2 * pitch_dw;
        lds = (size_t)tt * np_ * a.tcp * sizeof(uint16_t);
    }
    const uint64_t ntt = (rows + tt - 1) / tt, nct = (nc + tc - 1) / tc;
    if (ntt > 0xffffffffull || nct > 0xffffffffull) return BB_ERANGE;
    a.ntt = (uint32_t)ntt;
    a.nct = (uint32_t)nct;
    if (lds > 64 * 1024) return BB_ENOTSUP;
    uint64_t blocks = (uint64_t)nframes * ntt * nct;
    a.perm = make_perm(blocks, (uint64_t)nframes * rows * rowlen * 4, BB_ORDER_TILED);
    if (tb > 0 && blocks > (uint64_t)tb) blocks = (uint64_t)tb;
    if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
    const dim3 grid((unsigned)blocks), block(BB_BLOCK);
#define BB_TL(L) with_nt(nt, [&](auto NT) { hipLaunchKernelGGL((k_decode_i8_tiled<L, decltype(NT)::value>), grid, block, lds, st, a); })
#define BB_TS(L) with_nt(nt, [&](auto NT) { hipLaunchKernelGGL((k_decode_i8_stage<L, decltype(NT)::value>), grid, block, lds, st, a); })
    switch (p->layout) {
        case BB_LAYOUT_GUPPI_CF: BB_TL(0); break;
        case BB_LAYOUT_MKBF:     if (stage) BB_TS(1); else BB_TL(1); break;
        default:                 if (stage) BB_TS(2); else BB_TL(2); break;
    }
#undef BB_TS
#undef BB_TL
    BB_NOTE("%s<%d,%s> grid %u tile %u times x %u chans lds %zu",
            (stage && p->layout != BB_LAYOUT_GUPPI_CF) ? "k_decode_i8_stage" : "k_decode_i8_tiled",
            p->layout, nt ? "nt" : "plain", grid.x, a.tt, a.tc, lds);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_encode_flat(const float *d_in, size_t nelem, int coder, int bps,
                   void *d_out, size_t out_nbytes, void *stream)
{
    if (!coder_supported(coder, bps)) return BB_ENOTSUP;
    if (nelem == 0) return BB_OK;
    if (!d_in || !d_out) return BB_EINVAL;
    if ((nelem & 3) || ((nelem * (size_t)bps) & 7)) return BB_EINVAL;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 3)) return BB_EINVAL;
    if (out_nbytes < nelem * (size_t)bps / 8) return BB_ERANGE;
    { const int rc_ = ensure_init(); if (rc_) return rc_; }
    const uint64_t nquad = nelem / 4;
    // 4-bit codes (16-bit stores) gain from two runs per wave, 5.57 -> 6.05 TB/s at
    // 32 GiB in; every other width loses 1-8 % (profiles/r03zj_exp_encode_runs.log)
    const int eknob = g_tune_encode_runs.load();
    const int eruns = eknob ? eknob : (bps == 4 ? 2 : 1);
    uint64_t blocks = (nquad / 256 / eruns + 3) / 4 + 1;  // RUNS 256-quad runs per wave
    // one run per wave and as many workgroups as that takes: the encoder is a
    // streaming read without a software pipeline, the dispatcher overlaps it
    // best (profiles/r01g_exp_encode_grid.log: 4.7 -> 5.6 TB/s against 4096)
    const int tbe = g_tune_blocks.load();
    const uint64_t ecap = tbe > 0 ? (uint64_t)tbe : 0x7fffffffull;
    if (blocks > ecap) blocks = ecap;
    if (blocks == 0) blocks = 1;
    const dim3 grid((unsigned)blocks), block(BB_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    uint8_t *o = (uint8_t *)d_out;
    const bool direct = g_tune_encode_direct.load() != 0;
    bb_perm_t perm = {0, 0, 0};
    const int elw = g_tune_encode_lw.load();
    if (elw > 0 && ((nquad >> 8) >> elw) >= 64) {
        perm.lw = (uint32_t)elw;
        perm.stripe = (nquad >> 8) >> elw;
        perm.n = perm.stripe << elw;
    }
#define BB_E(C, B) launch_encode_flat<C, B>(direct, eruns, grid, block, st, d_in, nquad, o, perm)
    if (coder == BB_CODER_VDIF) {
        switch (bps) { case 1: BB_E(BB_CODER_VDIF, 1); break; case 2: BB_E(BB_CODER_VDIF, 2); break;
                       case 4: BB_E(BB_CODER_VDIF, 4); break; default: BB_E(BB_CODER_VDIF, 8); break; }
    } else if (coder == BB_CODER_MARK5B) {
        if (bps == 1) BB_E(BB_CODER_MARK5B, 1); else BB_E(BB_CODER_MARK5B, 2);
    } else {
        if (bps == 4) BB_E(BB_CODER_INT, 4); else BB_E(BB_CODER_INT, 8);
    }
#undef BB_E
    BB_NOTE("k_encode_flat<%s,%d,%s,%d> grid %u", coder == BB_CODER_VDIF ? "VDIF" : coder == BB_CODER_MARK5B ? "MARK5B" : "INT",
            bps, direct && bps == 2 ? "direct" : "thresholds", eruns, grid.x);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

int bb_encode_mark4(const float *d_in, size_t nwords, int ntrack,
                    const uint8_t sign_bit[32], const uint8_t mag_bit[32],
                    void *d_out, size_t out_nbytes, void *stream)
{
    if (ntrack != 16 && ntrack != 32 && ntrack != 64) return BB_ENOTSUP;
    if (nwords == 0) return BB_OK;
    if (!d_in || !d_out || !sign_bit || !mag_bit) return BB_EINVAL;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 7)) return BB_EINVAL;
    if (out_nbytes < nwords * (size_t)(ntrack / 8)) return BB_ERANGE;
    const int opw = ntrack / 2;
    for (int j = 0; j < opw; ++j)
        if (sign_bit[j] >= ntrack || mag_bit[j] >= ntrack) return BB_EINVAL;
    bb_m4enc_args a;
    a.in = d_in;
    a.out = (uint8_t *)d_out;
    a.nwords = nwords;
    memset(a.sign_bit, 0, sizeof(a.sign_bit));
    memset(a.mag_bit, 0, sizeof(a.mag_bit));
    memcpy(a.sign_bit, sign_bit, opw);
    memcpy(a.mag_bit, mag_bit, opw);
    const uint64_t nquad = (uint64_t)nwords * (ntrack / 8);
    // four quads per lane (measured optimum, profiles/r01g_exp_encode_grid.log)
    uint64_t blocks = (nquad + 4 * BB_BLOCK - 1) / (4 * BB_BLOCK);
    const int tbm = g_tune_blocks.load();
    const uint64_t mcap = tbm > 0 ? (uint64_t)tbm : 0x7fffffffull;
    if (blocks > mcap) blocks = mcap;
    if (blocks == 0) blocks = 1;
    const dim3 grid((unsigned)blocks), block(BB_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    { const int rc_ = ensure_init(); if (rc_) return rc_; }
    const bool direct = g_tune_encode_direct.load() != 0;
#define BB_M4E(N) do { if (direct) hipLaunchKernelGGL((k_encode_mark4<N, true>), grid, block, 0, st, a); \
                       else hipLaunchKernelGGL((k_encode_mark4<N, false>), grid, block, 0, st, a); } while (0)
    switch (ntrack) {
        case 16: BB_M4E(16); break;
        case 32: BB_M4E(32); break;
        default: BB_M4E(64); break;
    }
#undef BB_M4E
    BB_NOTE("k_encode_mark4<%d,%s> grid %u", ntrack, direct ? "direct" : "thresholds", grid.x);
    BB_HIP(hipGetLastError());
    return BB_OK;
}

} // extern "C"

#include "bbdecode_arena.h"
#include "bb_arena.inc"
