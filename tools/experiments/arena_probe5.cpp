// arena_probe5: is the rate of a mid-size launch a matter of WHEN it runs?
// arena_probe4 (profiles/r03d_arena_probe4.log) found 2^15- and 2^16-frame
// launches at 5.3 TB/s on every piece of HBM when nothing else happens between
// the measurements, 2^18-frame launches at 6.44 -- and arena_probe3 had the same
// chunks at 5.65-6.45 depending on what the driver was doing around them
// (clearing freshly created or released memory).  Here the SAME output buffer
// is decoded into in different temporal contexts:
//   isolated     launch, wait for it on the host, next launch (what the probes did)
//   back_to_back 48 launches queued at once, one event pair per launch
//   after_warm   a 25 ms fill of other memory queued right before every launch
//   with_fill    a slow background fill on a second stream while launches run
// Every launch takes the next window of an 8 GiB image (no input reuse).
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
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_fill(f4 *p, size_t n4, float v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) __builtin_nontemporal_store(f4{v, v, v, v}, &p[i]);
}

static const size_t FRAME = 8032, PAYLOAD = 8000, HDR = 32;
typedef std::chrono::steady_clock clk;
static uint8_t *g_in;
static const size_t IN_FRAMES = 1 << 20;
static size_t g_next = 0;

static int decode(float *out, size_t nframes, hipStream_t st)
{
    bb_decode_params p = {};
    p.coder = BB_CODER_VDIF; p.bps = 2; p.chunk = 1; p.nslot = 1;
    p.payload_nbytes = PAYLOAD; p.src0 = HDR; p.src_stride = FRAME;
    if (g_next + nframes > IN_FRAMES) g_next = 0;
    const size_t first = g_next; g_next += nframes;
    return bb_decode_frames(g_in + first * FRAME, nframes * FRAME, nullptr, nframes, &p, out, nframes * PAYLOAD * 4, st);
}

static double tbps(size_t nframes, float ms) { return (double)nframes * (FRAME + PAYLOAD * 16) / ms / 1e9; }

int main()
{
    CK(hipSetDevice(0));
    if (bb_init()) { fprintf(stderr, "bb_init failed\n"); return 1; }
    CK(hipMalloc((void **)&g_in, IN_FRAMES * FRAME + 256));
    hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, (uint32_t *)g_in, IN_FRAMES * FRAME / 4, 7u);
    CK(hipDeviceSynchronize());
    hipStream_t st, st2;
    CK(hipStreamCreate(&st)); CK(hipStreamCreate(&st2));
    const int N = 48;
    std::vector<hipEvent_t> ev(2 * N + 2);
    for (auto &e : ev) CK(hipEventCreate(&e));
    float *other; const size_t other_bytes = 40ull << 30;
    CK(hipMalloc(&other, other_bytes));

    for (size_t lf : {15, 16, 17, 18}) {
        const size_t nf = (size_t)1 << lf;
        for (int draw = 0; draw < 3; ++draw) {
            float *out; CK(hipMalloc(&out, nf * PAYLOAD * 16));
            // isolated
            std::vector<double> iso;
            for (int i = 0; i < 9; ++i) {
                CK(hipEventRecord(ev[0], st));
                if (decode(out, nf, st)) return 1;
                CK(hipEventRecord(ev[1], st)); CK(hipEventSynchronize(ev[1]));
                float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
                if (i) iso.push_back(tbps(nf, ms));
                struct timespec ts = {0, 2000000}; nanosleep(&ts, nullptr);         // 2 ms of idle
            }
            std::sort(iso.begin(), iso.end());
            // back to back
            CK(hipDeviceSynchronize());
            struct timespec ts = {0, 200000000}; nanosleep(&ts, nullptr);           // 0.2 s idle first
            for (int i = 0; i < N; ++i) {
                CK(hipEventRecord(ev[2 * i], st));
                if (decode(out, nf, st)) return 1;
                CK(hipEventRecord(ev[2 * i + 1], st));
            }
            CK(hipStreamSynchronize(st));
            printf("{\"log2_frames\": %zu, \"draw\": %d, \"isolated_median\": %.3f, \"isolated_min_max\": [%.3f, %.3f], \"back_to_back\": [", lf, draw,
                   iso[iso.size() / 2], iso.front(), iso.back());
            float tot; CK(hipEventElapsedTime(&tot, ev[0], ev[2 * N - 1]));
            for (int i = 0; i < N; ++i) {
                float ms; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
                printf("%s%.2f", i ? ", " : "", tbps(nf, ms));
            }
            printf("], \"back_to_back_whole_sequence\": %.3f", tbps(nf * N, tot));
            // after a warming fill
            CK(hipDeviceSynchronize());
            nanosleep(&ts, nullptr);
            std::vector<double> warm;
            for (int i = 0; i < 6; ++i) {
                hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, st, (f4 *)other, other_bytes / 16, 1.0f);     // ~6 ms at 6.9 TB/s per 40 GiB... several ms
                CK(hipEventRecord(ev[0], st));
                if (decode(out, nf, st)) return 1;
                CK(hipEventRecord(ev[1], st)); CK(hipEventSynchronize(ev[1]));
                float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
                warm.push_back(tbps(nf, ms));
                nanosleep(&ts, nullptr);
            }
            printf(", \"right_after_a_40GiB_fill\": [");
            for (size_t i = 0; i < warm.size(); ++i) printf("%s%.2f", i ? ", " : "", warm[i]);
            printf("]}\n");
            fflush(stdout);
            CK(hipFree(out));
        }
    }
    return 0;
}
