// arena_probe4: the HBM map again, cleanly.  Follow-up to arena_probe3
// (profiles/r03c_arena_probe3.log), where the same chunks decoded at 6.4, 5.65
// or 5.85 TB/s depending on WHEN they were measured and on whether the rest of
// memory was allocated.  Changes: (1) the input is an 8 GiB image and every
// launch decodes the NEXT window of it, so nothing of the input can stay in the
// 256 MiB Infinity Cache from one launch to the next (the earlier probes decoded
// the same 0.26-2.1 GB again and again); a second figure repeats one window, to
// size that effect; (2) free memory is covered by 1 GiB handles once and outputs
// are mapped from those handles directly -- no allocation happens between
// measurements, the set of buffer objects never changes until the last part.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include "bbdecode.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_rand(uint32_t *p, size_t n, uint32_t seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}

static const size_t FRAME = 8032, PAYLOAD = 8000, HDR = 32;
static const size_t GIB = 1ull << 30;
typedef std::chrono::steady_clock clk;
static clk::time_point T0;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
static double now_s() { return ms_since(T0) / 1e3; }
static hipEvent_t e0, e1;
static uint8_t *g_in;
static const size_t IN_FRAMES = 1 << 20;             // 8 GiB image
static size_t g_next = 0;                            // next window of the image
typedef hipMemGenericAllocationHandle_t handle_t;
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static char *va;

struct rates { double rotating, same; };

// median TB/s of `reps` launches; rotating: every launch takes the next nframes of the image
static double run(float *out, size_t nframes, bool rotate, int reps = 8)
{
    bb_decode_params p = {};
    p.coder = BB_CODER_VDIF; p.bps = 2; p.chunk = 1; p.nslot = 1;
    p.payload_nbytes = PAYLOAD; p.src0 = HDR; p.src_stride = FRAME;
    std::vector<double> t;
    for (int r = 0; r <= reps; ++r) {
        size_t first = 0;
        if (rotate) { if (g_next + nframes > IN_FRAMES) g_next = 0; first = g_next; g_next += nframes; }
        CK(hipEventRecord(e0));
        int rc = bb_decode_frames(g_in + first * FRAME, nframes * FRAME, nullptr, nframes, &p, out, nframes * PAYLOAD * 4, nullptr);
        if (rc) { fprintf(stderr, "bb_decode_frames rc %d\n", rc); exit(1); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return (double)nframes * (FRAME + PAYLOAD * 16) / t[t.size() / 2] / 1e9;
}

static std::vector<handle_t> sp;

static rates on_spacers(const std::vector<size_t> &idx, size_t nframes, bool both = true)
{
    for (size_t k = 0; k < idx.size(); ++k) CK(hipMemMap(va + k * GIB, GIB, 0, sp[idx[k]], 0));
    CK(hipMemSetAccess(va, idx.size() * GIB, &acc, 1));
    rates r;
    r.rotating = run((float *)va, nframes, true);
    r.same = both ? run((float *)va, nframes, false) : 0;
    CK(hipMemUnmap(va, idx.size() * GIB));
    return r;
}

static std::vector<size_t> seq(size_t first, size_t n, size_t step = 1)
{
    std::vector<size_t> v;
    for (size_t i = 0; i < n; ++i) v.push_back(first + i * step);
    return v;
}

int main()
{
    T0 = clk::now();
    CK(hipSetDevice(0));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (bb_init()) { fprintf(stderr, "bb_init failed\n"); return 1; }
    CK(hipMalloc((void **)&g_in, IN_FRAMES * FRAME + 256));
    hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, (uint32_t *)g_in, IN_FRAMES * FRAME / 4, 7u);
    CK(hipDeviceSynchronize());
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemAddressReserve((void **)&va, 40 * GIB, 0, nullptr, 0));
    const size_t NF16 = 1 << 16, NF18 = 1 << 18, NF15 = 1 << 15;

    // A. plain allocations, a fresh one per draw
    for (size_t nf : {NF15, NF16, NF18})
        for (int d = 0; d < 5; ++d) {
            float *a; CK(hipMalloc(&a, nf * PAYLOAD * 16));
            const double rr = run(a, nf, true), rs = run(a, nf, false);
            printf("{\"t_s\": %.1f, \"case\": \"hipMalloc\", \"frames\": %zu, \"draw\": %d, \"rotating_input\": %.3f, \"same_input\": %.3f}\n", now_s(), nf, d, rr, rs);
            fflush(stdout);
            CK(hipFree(a));
        }

    // B. cover free memory with 1 GiB handles (creation order = the driver's placement order)
    auto t0 = clk::now();
    for (;;) {
        handle_t h;
        if (hipMemCreate(&h, GIB, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        sp.push_back(h);
    }
    const size_t NS = sp.size();
    printf("{\"t_s\": %.1f, \"spacers_1GiB\": %zu, \"cover_ms\": %.0f}\n", now_s(), NS, ms_since(t0));

    // map: 8 adjacent handles (8 GiB) hold one 2^16-frame output; three passes
    for (int pass = 0; pass < 3; ++pass) {
        printf("{\"t_s\": %.1f, \"map_pass\": %d, \"frames\": %zu, \"unit\": \"8 adjacent 1 GiB handles\", \"rotating_input\": [", now_s(), pass, NF16);
        std::vector<double> same;
        for (size_t r0 = 0; r0 + 8 <= NS; r0 += 8) {
            const rates r = on_spacers(seq(r0, 8), NF16, pass == 0);
            same.push_back(r.same);
            printf("%s%.2f", r0 ? ", " : "", r.rotating);
            fflush(stdout);
        }
        printf("]");
        if (pass == 0) { printf(", \"same_input\": ["); for (size_t i = 0; i < same.size(); ++i) printf("%s%.2f", i ? ", " : "", same[i]); printf("]"); }
        printf("}\n");
        if (pass == 1) { struct timespec ts = {5, 0}; nanosleep(&ts, nullptr); }
    }
    // the same with 4 adjacent handles and 2^15 frames
    printf("{\"t_s\": %.1f, \"map\": \"4 adjacent handles\", \"frames\": %zu, \"rotating_input\": [", now_s(), NF15);
    for (size_t r0 = 0; r0 + 4 <= NS; r0 += 4) printf("%s%.2f", r0 ? ", " : "", on_spacers(seq(r0, 4), NF15, false).rotating);
    printf("]}\n");
    fflush(stdout);
    // and 32 adjacent handles, 2^18 frames
    printf("{\"t_s\": %.1f, \"map\": \"32 adjacent handles\", \"frames\": %zu, \"rotating_input\": [", now_s(), NF18);
    for (size_t r0 = 0; r0 + 32 <= NS; r0 += 16) printf("%s%.2f", r0 ? ", " : "", on_spacers(seq(r0, 32), NF18, false).rotating);
    printf("]}\n");
    fflush(stdout);

    // C. outputs made of handles from different places: 2^16 frames on 8 handles taken every `step`-th
    for (size_t step : {1, 2, 4, 8, 12, 16, 24, 32}) {
        if (7 * step >= NS) continue;
        printf("{\"t_s\": %.1f, \"frames\": %zu, \"handles\": \"8, every %zu-th\", \"rotating_input_by_first_handle\": {", now_s(), NF16, step);
        bool first = true;
        for (size_t f0 = 0; f0 + 7 * step < NS; f0 += std::max<size_t>(NS / 6, 1)) {
            printf("%s\"%zu\": %.2f", first ? "" : ", ", f0, on_spacers(seq(f0, 8, step), NF16, false).rotating);
            first = false;
        }
        printf("}}\n");
        fflush(stdout);
    }
    // 2^18 frames on 32 handles taken every step-th
    for (size_t step : {1, 2, 4, 8}) {
        if (31 * step >= NS) continue;
        printf("{\"t_s\": %.1f, \"frames\": %zu, \"handles\": \"32, every %zu-th\", \"rotating_input_by_first_handle\": {", now_s(), NF18, step);
        bool first = true;
        for (size_t f0 = 0; f0 + 31 * step < NS; f0 += std::max<size_t>(NS / 5, 1)) {
            printf("%s\"%zu\": %.2f", first ? "" : ", ", f0, on_spacers(seq(f0, 32, step), NF18, false).rotating);
            first = false;
        }
        printf("}}\n");
        fflush(stdout);
    }

    // D. the rest of memory: release every handle but the 8 spread ones and the 8 first ones, measure again
    {
        std::vector<size_t> spread = seq(0, 8, NS / 8), firsts = seq(1, 8);
        // (firsts starts at 1 so that handle 0 belongs to `spread` only)
        const rates a0 = on_spacers(spread, NF16), b0 = on_spacers(firsts, NF16);
        t0 = clk::now();
        for (size_t i = 0; i < NS; ++i) {
            if (std::find(spread.begin(), spread.end(), i) != spread.end()) continue;
            if (std::find(firsts.begin(), firsts.end(), i) != firsts.end()) continue;
            CK(hipMemRelease(sp[i]));
        }
        const double rel_ms = ms_since(t0);
        for (int round = 0; round < 3; ++round) {
            const rates a1 = on_spacers(spread, NF16), b1 = on_spacers(firsts, NF16);
            printf("{\"t_s\": %.1f, \"frames\": %zu, \"release_ms\": %.0f, \"spread8\": {\"rest_allocated\": [%.3f, %.3f], \"rest_free\": [%.3f, %.3f]}, "
                   "\"first8\": {\"rest_allocated\": [%.3f, %.3f], \"rest_free\": [%.3f, %.3f]}, \"pairs_are\": \"[rotating input, same input]\"}\n",
                   now_s(), NF16, rel_ms, a0.rotating, a0.same, a1.rotating, a1.same, b0.rotating, b0.same, b1.rotating, b1.same);
            fflush(stdout);
            struct timespec ts = {2, 0}; nanosleep(&ts, nullptr);
        }
        for (int d = 0; d < 3; ++d) {
            float *a;
            if (hipMalloc(&a, NF16 * PAYLOAD * 16) != hipSuccess) { (void)hipGetLastError(); continue; }
            printf("{\"t_s\": %.1f, \"case\": \"hipMalloc afterwards\", \"frames\": %zu, \"rotating_input\": %.3f, \"same_input\": %.3f}\n", now_s(), NF16, run(a, NF16, true), run(a, NF16, false));
            CK(hipFree(a));
        }
    }
    return 0;
}
