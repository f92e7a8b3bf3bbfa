// arena_probe3: a MAP of HBM.  Free memory is covered with 1 GiB spacer handles
// (hipMemCreate, creation order = the driver's placement order); a region of R
// adjacent spacers is then released and refilled with 2 MiB chunks, which can
// only land in that hole, and the cfg2 decode is timed on an output mapped from
// those chunks.  Gives: rate vs position of the region; whether a region's rate
// is stable over time; rates of outputs that combine two regions; and of an
// arena whose teeth are spread evenly, with the spacers held and released.
// Follow-up to profiles/r03a_arena_probe.log and r03b_arena_probe2.log.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <time.h>
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
static const size_t CHUNK = 2u << 20, SPACER = 1ull << 30, CPS = SPACER / CHUNK;
typedef std::chrono::steady_clock clk;
static clk::time_point T0;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
static hipEvent_t e0, e1;
static void *g_in;
typedef hipMemGenericAllocationHandle_t handle_t;
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static char *va;

static double decode_rate(float *out, size_t nframes, int reps = 4)
{
    bb_decode_params p = {};
    p.coder = BB_CODER_VDIF; p.bps = 2; p.chunk = 1; p.nslot = 1;
    p.payload_nbytes = PAYLOAD; p.src0 = HDR; p.src_stride = FRAME;
    std::vector<double> t;
    for (int r = 0; r <= reps; ++r) {
        CK(hipEventRecord(e0));
        int rc = bb_decode_frames(g_in, nframes * FRAME, nullptr, nframes, &p, out, nframes * PAYLOAD * 4, nullptr);
        if (rc) { fprintf(stderr, "bb_decode_frames rc %d\n", rc); exit(1); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return (double)nframes * (FRAME + PAYLOAD * 16) / t[t.size() / 2] / 1e9;   // TB/s
}

static double rate_of(const std::vector<handle_t> &l, size_t nframes)
{
    for (size_t k = 0; k < l.size(); ++k) CK(hipMemMap(va + k * CHUNK, CHUNK, 0, l[k], 0));
    CK(hipMemSetAccess(va, l.size() * CHUNK, &acc, 1));
    const double r = decode_rate((float *)va, nframes);
    CK(hipMemUnmap(va, l.size() * CHUNK));
    return r;
}

static std::vector<handle_t> sp;          // spacers, creation order
static std::vector<char> sp_live;

static void open_hole(size_t i) { if (sp_live[i]) { CK(hipMemRelease(sp[i])); sp_live[i] = 0; } }
// (released memory comes back to the allocator with a delay: retry for a while)
static double g_wait_ms = 0;
static void close_hole(size_t i)
{
    if (sp_live[i]) return;
    auto t0 = clk::now();
    for (int tries = 0; tries < 20000; ++tries) {
        if (hipMemCreate(&sp[i], SPACER, &prop, 0) == hipSuccess) { sp_live[i] = 1; g_wait_ms += ms_since(t0); return; }
        (void)hipGetLastError();
        struct timespec ts = {0, 500000}; nanosleep(&ts, nullptr);
    }
    fprintf(stderr, "close_hole(%zu): memory did not come back\n", i); exit(1);
}
static std::vector<handle_t> fill(size_t nchunks)
{
    std::vector<handle_t> h(nchunks);
    for (size_t i = 0; i < nchunks; ++i) {
        int tries = 0;
        while (hipMemCreate(&h[i], CHUNK, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (++tries > 20000) { fprintf(stderr, "fill: out of memory at chunk %zu\n", i); exit(1); }
            struct timespec ts = {0, 500000}; nanosleep(&ts, nullptr);
        }
    }
    return h;
}
static void release(std::vector<handle_t> &h) { for (auto x : h) CK(hipMemRelease(x)); h.clear(); }
static std::vector<handle_t> first_n(const std::vector<handle_t> &p, size_t n) { return std::vector<handle_t>(p.begin(), p.begin() + n); }
static std::vector<handle_t> dealt(const std::vector<handle_t> &p, size_t need, size_t T)
{
    std::vector<handle_t> l; std::vector<size_t> used(T, 0);
    for (size_t k = 0; k < need; ++k) { const size_t t = k % T; l.push_back(p[t * p.size() / T + used[t]++]); }
    return l;
}
static size_t need_chunks(size_t nf) { return (nf * PAYLOAD * 16 + CHUNK - 1) / CHUNK; }

int main(int argc, char **argv)
{
    T0 = clk::now();
    CK(hipSetDevice(0));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (bb_init()) { fprintf(stderr, "bb_init failed\n"); return 1; }
    const size_t max_frames = 1 << 17;
    CK(hipMalloc(&g_in, max_frames * FRAME + 256));
    hipLaunchKernelGGL(k_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)g_in, max_frames * FRAME / 4, 7u);
    CK(hipDeviceSynchronize());
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemAddressReserve((void **)&va, (max_frames * PAYLOAD * 16 / CHUNK + 64) * CHUNK, 0, nullptr, 0));

    for (int r = 0; r < 3; ++r) {
        const size_t nf = 1 << 16;
        float *a; CK(hipMalloc(&a, nf * PAYLOAD * 16));
        printf("{\"t_s\": %.1f, \"case\": \"hipMalloc\", \"log2_frames\": 16, \"TBps\": %.3f}\n", ms_since(T0) / 1e3, decode_rate(a, nf));
        CK(hipFree(a));
    }
    // cover free memory: 1 GiB spacers, then 64 MiB pieces for the remainder
    auto t0 = clk::now();
    for (;;) {
        handle_t h;
        if (hipMemCreate(&h, SPACER, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        sp.push_back(h);
    }
    sp_live.assign(sp.size(), 1);
    std::vector<handle_t> crumbs;
    for (;;) {
        handle_t h;
        if (hipMemCreate(&h, 64u << 20, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        crumbs.push_back(h);
    }
    size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
    printf("{\"t_s\": %.1f, \"spacers\": %zu, \"crumbs_64MiB\": %zu, \"cover_ms\": %.1f, \"free_after_MiB\": %.0f}\n",
           ms_since(T0) / 1e3, sp.size(), crumbs.size(), ms_since(t0), free_b / 1048576.0);
    fflush(stdout);
    const size_t NS = sp.size();

    // 1. map: region r = 8 adjacent spacers (8 GiB) released and refilled with 2 MiB chunks,
    //    which stay (they hold the place from then on); 2^16 frames on each region
    const size_t R = 8, NR = NS / R;
    std::vector<std::vector<handle_t>> grp(NR);
    std::vector<double> region_rate;
    printf("{\"t_s\": %.1f, \"map_pass\": 0, \"region_GiB\": %zu, \"log2_frames\": 16, \"TBps\": [", ms_since(T0) / 1e3, R);
    for (size_t r = 0; r < NR; ++r) {
        for (size_t i = r * R; i < (r + 1) * R; ++i) open_hole(i);
        grp[r] = fill(R * CPS);
        const double rt = rate_of(first_n(grp[r], need_chunks(1 << 16)), 1 << 16);
        region_rate.push_back(rt);
        printf("%s%.2f", r ? ", " : "", rt);
        fflush(stdout);
    }
    printf("]}\n");
    for (int pass = 1; pass < 3; ++pass) {
        printf("{\"t_s\": %.1f, \"map_pass\": %d, \"region_GiB\": %zu, \"log2_frames\": 16, \"TBps\": [", ms_since(T0) / 1e3, pass, R);
        for (size_t r = 0; r < NR; ++r) printf("%s%.2f", r ? ", " : "", rate_of(first_n(grp[r], need_chunks(1 << 16)), 1 << 16));
        printf("]}\n");
        fflush(stdout);
    }
    printf("{\"t_s\": %.1f, \"map_region_GiB\": 4, \"log2_frames\": 15, \"TBps\": [", ms_since(T0) / 1e3);
    for (size_t r = 0; r < NR; ++r)
        for (size_t half = 0; half < 2; ++half) {
            std::vector<handle_t> l(grp[r].begin() + half * 2048, grp[r].begin() + half * 2048 + need_chunks(1 << 15));
            printf("%s%.2f", (r || half) ? ", " : "", rate_of(l, 1 << 15));
        }
    printf("]}\n");
    fflush(stdout);
    // 2. two regions at a time (4 GiB of each): one after the other / alternating chunks
    {
        std::vector<size_t> order(NR);
        for (size_t i = 0; i < NR; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return region_rate[a] < region_rate[b]; });
        const size_t S0 = order[0], S1 = order[1], F0 = order[NR - 1], F1 = order[NR - 2], M = order[NR / 2];
        const size_t pairs[][2] = {{S0, S1}, {S0, F0}, {F0, F1}, {S0, M}, {F0, M}, {S1, F1}, {0, NR - 1}, {0, NR / 2}};
        for (auto &pr : pairs) {
            std::vector<handle_t> pool(grp[pr[0]].begin(), grp[pr[0]].begin() + 2048);
            pool.insert(pool.end(), grp[pr[1]].begin(), grp[pr[1]].begin() + 2048);
            const size_t need = need_chunks(1 << 16);
            printf("{\"t_s\": %.1f, \"pair_regions\": [%zu, %zu], \"alone\": [%.2f, %.2f], \"log2_frames\": 16, \"halves_in_turn\": %.3f, \"alternating_chunks\": %.3f}\n",
                   ms_since(T0) / 1e3, pr[0], pr[1], region_rate[pr[0]], region_rate[pr[1]],
                   rate_of(first_n(pool, need), 1 << 16), rate_of(dealt(pool, need, 2), 1 << 16));
            fflush(stdout);
        }
    }
    // 3. arenas of 32 GiB: 32 teeth of 1 GiB (512 chunks) spread evenly over the regions / adjacent
    auto arena = [&](int kind) {
        std::vector<handle_t> pool;
        for (size_t t = 0; t < 32; ++t) {
            // tooth t = GiB number g of the map (region g / 8, eighth g % 8)
            const size_t g = kind == 0 ? t * (NR * R) / 32 : kind == 1 ? t : kind == 2 ? NR * R / 2 + t : NR * R - 32 + t;
            const auto &v = grp[g / R];
            pool.insert(pool.end(), v.begin() + (g % R) * CPS, v.begin() + (g % R + 1) * CPS);
        }
        return pool;
    };
    const char *names[] = {"spread", "first32", "middle32", "last32"};
    for (int kind = 0; kind < 4; ++kind) {
        std::vector<handle_t> pool = arena(kind);
        for (size_t lf : {15, 16, 17})
            for (int rep = 0; rep < 2; ++rep) {
                const size_t nf = (size_t)1 << lf, need = need_chunks(nf);
                printf("{\"t_s\": %.1f, \"arena\": \"%s\", \"rest_of_memory\": \"allocated\", \"log2_frames\": %zu, \"consecutive\": %.3f, \"dealt32\": %.3f, \"dealt512\": %.3f}\n",
                       ms_since(T0) / 1e3, names[kind], lf, rate_of(first_n(pool, need), nf), rate_of(dealt(pool, need, 32), nf),
                       rate_of(dealt(pool, need, 512), nf));
                fflush(stdout);
            }
    }
    // 4. keep only the spread arena: everything else goes back to the driver
    {
        std::vector<handle_t> pool = arena(0);
        std::vector<char> keep_mark;
        auto t0r = clk::now();
        for (size_t r = 0; r < NR; ++r)
            for (size_t c = 0; c < grp[r].size(); ++c) {
                const size_t g = r * R + c / CPS;
                bool keep = false;
                for (size_t t = 0; t < 32; ++t) if (t * (NR * R) / 32 == g) keep = true;
                if (!keep) CK(hipMemRelease(grp[r][c]));
            }
        for (size_t i = 0; i < NS; ++i) open_hole(i);
        for (auto c : crumbs) CK(hipMemRelease(c));
        printf("{\"t_s\": %.1f, \"released_all_but_spread_arena_ms\": %.0f}\n", ms_since(T0) / 1e3, ms_since(t0r));
        for (int round = 0; round < 2; ++round) {
            for (size_t lf : {15, 16, 17}) {
                const size_t nf = (size_t)1 << lf, need = need_chunks(nf);
                printf("{\"t_s\": %.1f, \"arena\": \"spread\", \"rest_of_memory\": \"free\", \"log2_frames\": %zu, \"consecutive\": %.3f, \"dealt32\": %.3f, \"dealt512\": %.3f}\n",
                       ms_since(T0) / 1e3, lf, rate_of(first_n(pool, need), nf), rate_of(dealt(pool, need, 32), nf), rate_of(dealt(pool, need, 512), nf));
                fflush(stdout);
            }
            for (int r = 0; r < 2; ++r) {
                const size_t nf = 1 << 16;
                float *a;
                if (hipMalloc(&a, nf * PAYLOAD * 16) != hipSuccess) { (void)hipGetLastError(); printf("{\"hipMalloc beside the arena\": \"failed\"}\n"); continue; }
                printf("{\"t_s\": %.1f, \"case\": \"hipMalloc beside the spread arena\", \"log2_frames\": 16, \"TBps\": %.3f}\n", ms_since(T0) / 1e3, decode_rate(a, nf));
                CK(hipFree(a));
            }
            struct timespec ts = {3, 0}; nanosleep(&ts, nullptr);
        }
    }
    return 0;
}
