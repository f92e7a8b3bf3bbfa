// grow_probe: what does taking device memory cost once it has been used before
// (the 1.5-4.6 s of a first large read), by which call, and
// can it be spread over threads or hidden behind other GPU work?
//
//   1. make every byte "recycled": hipMalloc most of the device, touch it, free it
//   2. time hipMalloc(48 GiB)                         (what torch.empty pays)
//   3. time 1536 x hipMemCreate(32 MiB), 1 thread     (the arena's step today)
//   4. the same on 2 / 4 / 8 / 16 threads
//   5. 48 x hipMemCreate(1 GiB); 192 x 256 MiB
//   6. while ONE background thread creates 48 GiB: the main thread launches a
//      small kernel + hipMemcpyAsync H2D in a loop -- are they stalled?
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/grow_probe tools/experiments/grow_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void k_touch(uint32_t *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (uint32_t)i;
}

__global__ void k_small(float *p) { p[threadIdx.x] += 1.f; }

static hipMemAllocationProp g_prop;

// create `total` bytes in chunks of `chunk` on `nthreads` threads; returns ms; handles appended
static double create_all(size_t total, size_t chunk, int nthreads, std::vector<hipMemGenericAllocationHandle_t> &out)
{
    const size_t n = total / chunk;
    out.assign(n, 0);
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    const double t0 = now_ms();
    auto work = [&]() {
        (void)hipSetDevice(0);
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n) break;
            if (hipMemCreate(&out[k], chunk, &g_prop, 0) != hipSuccess) { (void)hipGetLastError(); failed++; out[k] = 0; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    const double ms = now_ms() - t0;
    if (failed) printf("   (%d creates FAILED)\n", (int)failed);
    return ms;
}

static void release_all(std::vector<hipMemGenericAllocationHandle_t> &h)
{
    for (auto x : h) if (x) (void)hipMemRelease(x);
    h.clear();
}

int main(int argc, char **argv)
{
    const size_t GiB = 1ull << 30;
    const size_t step = (argc > 1 ? (size_t)atol(argv[1]) : 48) * GiB;
    CK(hipSetDevice(0));
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("device: %.1f GiB free of %.1f\n", free_b / 1073741824.0, total_b / 1073741824.0);

    if (argc > 2 && atoi(argv[2]) == 2) {
        // MODE 2: is the cost a property of the memory ("recycled") or of a release that
        // has not finished yet?  free -> create at once / after a pause.
        auto cycle = [&](size_t big_gib, double pause_s, const char *how) {
            void *p = nullptr;
            CK(hipMalloc(&p, big_gib * GiB));
            k_touch<<<65536, 256>>>((uint32_t *)p, big_gib * GiB / 4);
            CK(hipDeviceSynchronize());
            double t0 = now_ms();
            CK(hipFree(p));
            const double t_free = now_ms() - t0;
            if (pause_s > 0) std::this_thread::sleep_for(std::chrono::duration<double>(pause_s));
            std::vector<hipMemGenericAllocationHandle_t> h;
            const double t_create = create_all(step, 32u << 20, 1, h);
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, step, 32u << 20, nullptr, 0));
            t0 = now_ms();
            for (size_t k = 0; k < h.size(); ++k) CK(hipMemMap((char *)va + k * (32u << 20), 32u << 20, 0, h[k], 0));
            hipMemAccessDesc acc = {};
            acc.location = g_prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, step, &acc, 1));
            const double t_map = now_ms() - t0;
            t0 = now_ms();
            k_touch<<<65536, 256>>>((uint32_t *)va, step / 4);
            CK(hipDeviceSynchronize());
            const double t_touch = now_ms() - t0;
            t0 = now_ms();
            for (size_t k = 0; k < h.size(); ++k) (void)hipMemUnmap((char *)va + k * (32u << 20), 32u << 20);
            const double t_unmap = now_ms() - t0;
            t0 = now_ms();
            release_all(h);
            const double t_rel = now_ms() - t0;
            CK(hipMemAddressFree(va, step));
            printf("M2 %s: hipFree(%zu GiB) %.1f ms, pause %.1f s, create %zu GiB %.1f ms, map %.1f ms, first touch %.1f ms, "
                   "unmap %.1f ms, release %.1f ms\n", how, big_gib, t_free, pause_s, step / GiB, t_create, t_map, t_touch,
                   t_unmap, t_rel);
            fflush(stdout);
        };
        cycle(128, 0, "free -> create at once     ");
        std::this_thread::sleep_for(std::chrono::seconds(8));
        cycle(128, 8, "free -> 8 s -> create      ");
        std::this_thread::sleep_for(std::chrono::seconds(8));
        cycle(128, 2, "free -> 2 s -> create      ");
        std::this_thread::sleep_for(std::chrono::seconds(8));
        cycle(32, 0, "free 32 GiB -> create 48   ");
        std::this_thread::sleep_for(std::chrono::seconds(8));
        // release of VMM handles, then create again at once / after a pause
        for (double pause : {0.0, 3.0, 0.0}) {
            std::vector<hipMemGenericAllocationHandle_t> h;
            const double c1 = create_all(step, 32u << 20, 1, h);
            double t0 = now_ms();
            release_all(h);
            const double r1 = now_ms() - t0;
            if (pause > 0) std::this_thread::sleep_for(std::chrono::duration<double>(pause));
            const double c2 = create_all(step, 32u << 20, 1, h);
            t0 = now_ms();
            release_all(h);
            printf("M2 VMM: create %.1f ms, release %.1f ms, pause %.1f s, create again %.1f ms, release %.1f ms\n", c1, r1, pause,
                   c2, now_ms() - t0);
            fflush(stdout);
            std::this_thread::sleep_for(std::chrono::seconds(6));
        }
        return 0;
    }
    // 0. fresh memory first: what a step costs before anything was used
    {
        std::vector<hipMemGenericAllocationHandle_t> h;
        const double ms = create_all(step, 32u << 20, 1, h);
        printf("0. FRESH device: %zu x hipMemCreate(32 MiB), 1 thread: %.1f ms\n", h.size(), ms);
        release_all(h);
    }
    // 1. recycle: take most of the device, write it, give it back
    {
        const size_t big = (free_b - 8 * GiB) / GiB * GiB;
        void *p = nullptr;
        double t0 = now_ms();
        CK(hipMalloc(&p, big));
        const double t_alloc = now_ms() - t0;
        t0 = now_ms();
        k_touch<<<65536, 256>>>((uint32_t *)p, big / 4);
        CK(hipDeviceSynchronize());
        const double t_touch = now_ms() - t0;
        t0 = now_ms();
        CK(hipFree(p));
        printf("1. hipMalloc(%zu GiB) %.1f ms, touched in %.1f ms (%.0f GB/s), hipFree %.1f ms\n", big / GiB, t_alloc,
               t_touch, big / t_touch / 1e6, now_ms() - t0);
    }
    // 2. hipMalloc of one step, recycled memory (twice)
    for (int r = 0; r < 2; ++r) {
        void *p = nullptr;
        double t0 = now_ms();
        CK(hipMalloc(&p, step));
        const double t_alloc = now_ms() - t0;
        t0 = now_ms();
        k_touch<<<65536, 256>>>((uint32_t *)p, step / 4);
        CK(hipDeviceSynchronize());
        const double t_touch = now_ms() - t0;
        t0 = now_ms();
        CK(hipFree(p));
        printf("2. recycled: hipMalloc(%zu GiB) %.1f ms; first touch %.1f ms (%.0f GB/s); hipFree %.1f ms\n", step / GiB,
               t_alloc, t_touch, step / t_touch / 1e6, now_ms() - t0);
    }
    // 3 + 4. hipMemCreate in 32 MiB chunks on 1 .. 16 threads
    for (int nt : {1, 2, 4, 8, 16, 1}) {
        std::vector<hipMemGenericAllocationHandle_t> h;
        const double ms = create_all(step, 32u << 20, nt, h);
        double t0 = now_ms();
        release_all(h);
        printf("3. recycled: %zu x hipMemCreate(32 MiB) on %2d thread(s): %.1f ms (%.1f GB/s); release %.1f ms\n",
               step / (32u << 20), nt, ms, step / ms / 1e6, now_ms() - t0);
    }
    // 5. chunk sizes
    for (size_t chunk : {(size_t)256 << 20, (size_t)1 << 30, (size_t)2 << 20}) {
        std::vector<hipMemGenericAllocationHandle_t> h;
        const size_t tot = chunk == ((size_t)2 << 20) ? 8 * GiB : step;
        const double ms = create_all(tot, chunk, 1, h);
        release_all(h);
        printf("5. recycled: %zu x hipMemCreate(%zu MiB), 1 thread: %.1f ms (%.1f GB/s)\n", tot / chunk, chunk >> 20, ms,
               tot / ms / 1e6);
    }
    // 6. does a background creation stall launches and copies of the main thread?
    {
        float *d = nullptr;
        void *hbuf = nullptr, *dbuf = nullptr;
        const size_t cp = 64u << 20;
        CK(hipMalloc(&d, 4096));
        CK(hipMalloc(&dbuf, cp));
        CK(hipHostMalloc(&hbuf, cp, 0));
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        auto loop = [&](double for_ms, const char *what, std::atomic<bool> *until) {
            double worst_launch = 0, worst_copy = 0, sum_copy = 0;
            int n = 0;
            const double t_end = now_ms() + for_ms;
            while (until ? !until->load() : now_ms() < t_end) {
                double t0 = now_ms();
                k_small<<<1, 64, 0, s>>>(d);
                CK(hipStreamSynchronize(s));
                double dt = now_ms() - t0;
                worst_launch = dt > worst_launch ? dt : worst_launch;
                t0 = now_ms();
                CK(hipMemcpyAsync(dbuf, hbuf, cp, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
                dt = now_ms() - t0;
                worst_copy = dt > worst_copy ? dt : worst_copy;
                sum_copy += dt;
                ++n;
            }
            printf("6. %s: %d rounds, launch+sync worst %.3f ms, 64 MiB H2D mean %.3f ms (%.1f GB/s) worst %.3f ms\n", what, n,
                   worst_launch, sum_copy / n, cp / (sum_copy / n) / 1e6, worst_copy);
        };
        loop(300, "alone", nullptr);
        for (int nt : {1, 4}) {
            std::atomic<bool> done{false};
            std::vector<hipMemGenericAllocationHandle_t> h;
            double ms = 0;
            std::thread bg([&]() { ms = create_all(step, 32u << 20, nt, h); done = true; });
            char what[96];
            snprintf(what, sizeof what, "while %d background thread(s) create %zu GiB", nt, step / GiB);
            loop(0, what, &done);
            bg.join();
            printf("   (the background creation took %.1f ms)\n", ms);
            release_all(h);
        }
    }
    return 0;
}
