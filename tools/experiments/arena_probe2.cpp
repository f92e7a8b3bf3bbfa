// arena_probe2: follow-up to arena_probe (profiles/r03a_arena_probe.log), which
// showed: outputs mapped from 2 MiB hipMemCreate chunks never land in the slow
// mode of plain allocations (6.05-6.38 TB/s against 5.3-6.5), and chunks dealt
// round robin over >= 64 places spread over ALL of HBM reach 6.3-6.6.  An arena
// cannot hold all of HBM, so:
//   pool    chunks of ONE pool of P GiB created in one go: consecutive, dealt
//           over T teeth inside the pool, randomly permuted
//   sizes   the same pool idea with chunks of 512 KiB .. 32 MiB
//   steer   cover free HBM with 1 GiB spacer handles, release K of them evenly
//           spread and create the arena's 2 MiB chunks into the holes
// Build: make -C baseband_amd/csrc arena_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <random>
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
typedef std::chrono::steady_clock clk;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
static hipEvent_t e0, e1;
static void *g_in;
typedef hipMemGenericAllocationHandle_t handle_t;

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

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static char *va;

static double rate_of(const std::vector<handle_t> &l, size_t chunk, size_t nframes, double *map_ms = nullptr)
{
    auto t0 = clk::now();
    for (size_t k = 0; k < l.size(); ++k) CK(hipMemMap(va + k * chunk, chunk, 0, l[k], 0));
    CK(hipMemSetAccess(va, l.size() * chunk, &acc, 1));
    if (map_ms) *map_ms = ms_since(t0);
    const double r = decode_rate((float *)va, nframes);
    CK(hipMemUnmap(va, l.size() * chunk));
    return r;
}

static std::vector<handle_t> make_chunks(size_t n, size_t chunk, double *ms)
{
    std::vector<handle_t> h(n);
    auto t0 = clk::now();
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    if (ms) *ms = ms_since(t0);
    return h;
}

static void release(std::vector<handle_t> &h) { for (auto x : h) CK(hipMemRelease(x)); h.clear(); }

// chunks of `pool` dealt round robin over T teeth (tooth t = consecutive run starting at t * n / T)
static std::vector<handle_t> deal(const std::vector<handle_t> &pool, size_t need, size_t T)
{
    std::vector<handle_t> l;
    std::vector<size_t> used(T, 0);
    for (size_t k = 0; k < need; ++k) {
        const size_t t = k % T;
        l.push_back(pool[t * pool.size() / T + used[t]++]);
    }
    return l;
}

int main(int argc, char **argv)
{
    bool do_pool = false, do_sizes = false, do_steer = false, do_base = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "pool")) do_pool = true;
        else if (!strcmp(argv[i], "sizes")) do_sizes = true;
        else if (!strcmp(argv[i], "steer")) do_steer = true;
        else if (!strcmp(argv[i], "base")) do_base = true;
    }
    CK(hipSetDevice(0));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (bb_init()) { fprintf(stderr, "bb_init failed\n"); return 1; }
    const size_t max_frames = 1 << 18;
    CK(hipMalloc(&g_in, max_frames * FRAME + 256));
    hipLaunchKernelGGL(k_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)g_in, max_frames * FRAME / 4, 7u);
    CK(hipDeviceSynchronize());
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t va_bytes = (max_frames * PAYLOAD * 16 / (32u << 20) + 2) * (32u << 20);
    CK(hipMemAddressReserve((void **)&va, va_bytes, 0, nullptr, 0));

    if (do_base) {
        for (size_t lf : {15, 16, 18})
            for (int r = 0; r < 4; ++r) {
                const size_t nf = (size_t)1 << lf;
                float *a; CK(hipMalloc(&a, nf * PAYLOAD * 16));
                printf("{\"case\": \"hipMalloc\", \"log2_frames\": %zu, \"draw\": %d, \"TBps\": %.3f}\n", lf, r, decode_rate(a, nf));
                fflush(stdout);
                CK(hipFree(a));
            }
    }

    if (do_pool) {
        const size_t chunk = 2u << 20;
        for (size_t pool_gib : {10, 16, 40}) {
            double cms;
            std::vector<handle_t> pool = make_chunks(pool_gib * 512, chunk, &cms);
            printf("{\"pool_GiB\": %zu, \"chunks\": %zu, \"create_ms\": %.1f}\n", pool_gib, pool.size(), cms);
            for (size_t lf : {15, 16, 18}) {
                const size_t nf = (size_t)1 << lf;
                const size_t need = (nf * PAYLOAD * 16 + chunk - 1) / chunk;
                if (need > pool.size()) continue;
                double mms;
                std::vector<handle_t> l(pool.begin(), pool.begin() + need);
                printf("{\"pool_GiB\": %zu, \"log2_frames\": %zu, \"layout\": \"consecutive\", \"TBps\": %.3f", pool_gib, lf, rate_of(l, chunk, nf, &mms));
                printf(", \"map_ms\": %.1f}\n", mms);
                for (size_t T : {4, 16, 64, 256, 1024, 4096}) {
                    if (T > pool.size() || (need + T - 1) / T > pool.size() / T) continue;
                    for (int rep = 0; rep < 2; ++rep)
                        printf("{\"pool_GiB\": %zu, \"log2_frames\": %zu, \"layout\": \"teeth\", \"teeth\": %zu, \"TBps\": %.3f}\n",
                               pool_gib, lf, T, rate_of(deal(pool, need, T), chunk, nf));
                    fflush(stdout);
                }
                for (unsigned seed : {1u, 2u, 3u}) {
                    std::vector<handle_t> sh = pool;
                    std::mt19937 rng(seed);
                    std::shuffle(sh.begin(), sh.end(), rng);
                    sh.resize(need);
                    printf("{\"pool_GiB\": %zu, \"log2_frames\": %zu, \"layout\": \"random\", \"seed\": %u, \"TBps\": %.3f}\n",
                           pool_gib, lf, seed, rate_of(sh, chunk, nf));
                    fflush(stdout);
                }
            }
            release(pool);
        }
    }

    if (do_sizes) {
        const size_t pool_bytes = 16ull << 30;
        for (size_t kib : {512, 1024, 2048, 4096, 8192, 32768}) {
            const size_t chunk = kib << 10;
            double cms;
            std::vector<handle_t> pool = make_chunks(pool_bytes / chunk, chunk, &cms);
            for (size_t lf : {15, 16}) {
                const size_t nf = (size_t)1 << lf;
                const size_t need = (nf * PAYLOAD * 16 + chunk - 1) / chunk;
                double mms;
                std::vector<handle_t> l(pool.begin(), pool.begin() + need);
                const double rc = rate_of(l, chunk, nf, &mms);
                std::vector<handle_t> sh = pool;
                std::mt19937 rng(5);
                std::shuffle(sh.begin(), sh.end(), rng);
                sh.resize(need);
                const double rr = rate_of(sh, chunk, nf);
                const size_t T = pool.size() < 256 ? pool.size() / 2 : 256;
                const double rt = rate_of(deal(pool, need, T), chunk, nf);
                printf("{\"chunk_KiB\": %zu, \"log2_frames\": %zu, \"create_ms\": %.1f, \"map_ms\": %.1f, \"consecutive\": %.3f, \"random\": %.3f, \"teeth%zu\": %.3f}\n",
                       kib, lf, cms, mms, rc, rr, T, rt);
                fflush(stdout);
            }
            release(pool);
        }
    }

    if (do_steer) {
        // spacers over (nearly) all free memory, then chunks into evenly spread holes
        const size_t chunk = 2u << 20, spacer = 1ull << 30;
        size_t free_b = 0, total_b = 0;
        CK(hipMemGetInfo(&free_b, &total_b));
        for (size_t arena_gib : {16, 32}) {
            for (size_t K : {16, 64}) {
                CK(hipMemGetInfo(&free_b, &total_b));
                const size_t nsp = (free_b - (arena_gib + 4ull << 30)) / spacer;     // leave arena + 4 GiB free
                auto t0 = clk::now();
                std::vector<handle_t> sp;
                for (size_t i = 0; i < nsp; ++i) {
                    handle_t h;
                    if (hipMemCreate(&h, spacer, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
                    sp.push_back(h);
                }
                const double sp_ms = ms_since(t0);
                // the arena's free share first (so that the holes are the only other free memory):
                // take what is free now as `rest`, open the holes, fill them, then drop `rest`
                t0 = clk::now();
                std::vector<handle_t> rest;
                for (;;) {
                    handle_t h;
                    if (hipMemCreate(&h, 256u << 20, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
                    rest.push_back(h);
                    CK(hipMemGetInfo(&free_b, &total_b));
                    if (free_b < (3ull << 30)) break;
                }
                const double rest_ms = ms_since(t0);
                t0 = clk::now();
                const size_t per_tooth = arena_gib * 512 / K;              // chunks per tooth
                const size_t holes_per_tooth = (per_tooth * chunk + spacer - 1) / spacer;
                std::vector<handle_t> pool;
                std::vector<char> gone(sp.size(), 0);
                for (size_t t = 0; t < K; ++t) {
                    size_t s0 = t * sp.size() / K;
                    for (size_t j = 0; j < holes_per_tooth && s0 + j < sp.size(); ++j)
                        if (!gone[s0 + j]) { CK(hipMemRelease(sp[s0 + j])); gone[s0 + j] = 1; }
                    for (size_t c = 0; c < per_tooth; ++c) {
                        handle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
                        pool.push_back(h);
                    }
                }
                const double fill_ms = ms_since(t0);
                t0 = clk::now();
                for (size_t i = 0; i < sp.size(); ++i) if (!gone[i]) CK(hipMemRelease(sp[i]));
                release(rest);
                const double drop_ms = ms_since(t0);
                printf("{\"steer_arena_GiB\": %zu, \"teeth\": %zu, \"spacers\": %zu, \"spacer_ms\": %.1f, \"rest_ms\": %.1f, \"fill_ms\": %.1f, \"drop_ms\": %.1f}\n",
                       arena_gib, K, sp.size(), sp_ms, rest_ms, fill_ms, drop_ms);
                for (size_t lf : {15, 16, 17}) {
                    const size_t nf = (size_t)1 << lf;
                    const size_t need = (nf * PAYLOAD * 16 + chunk - 1) / chunk;
                    if (need > pool.size()) continue;
                    for (int rep = 0; rep < 2; ++rep) {
                        printf("{\"steer_arena_GiB\": %zu, \"teeth\": %zu, \"log2_frames\": %zu, \"dealt\": %.3f", arena_gib, K, lf,
                               rate_of(deal(pool, need, K), chunk, nf));
                        std::vector<handle_t> l(pool.begin(), pool.begin() + need);
                        printf(", \"consecutive\": %.3f}\n", rate_of(l, chunk, nf));
                        fflush(stdout);
                    }
                }
                release(pool);
            }
        }
    }
    return 0;
}
