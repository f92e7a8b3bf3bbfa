// vmm_probe: can a decode OUTPUT be laid over distant physical regions of HBM
// with the HIP virtual-memory API, and does that lift small launches the way a
// large launch's own span does?  ("Where the output lies";
// profiles/r02b_exp_stripe.log, r02c_exp_stripe2.log.)
//
// Output of a 2^16-frame cfg2 launch (8.4 GB) is placed
//   A  in one hipMalloc allocation,
//   B  in a VMM range whose chunks are created and mapped in order,
//   C  in a VMM range whose chunks come from R groups of physical memory that
//      were created with a large spacer allocation between them, mapped round
//      robin (chunk k <- group k % R).
// Build: make -C baseband_amd/csrc vmm_probe.   Usage: vmm_probe [chunk MiB] [spacer GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
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

static double decode_rate(const void *in, size_t in_bytes, float *out, size_t nframes)
{
    bb_decode_params p = {};
    p.coder = BB_CODER_VDIF; p.bps = 2; p.chunk = 1; p.nslot = 1;
    p.payload_nbytes = PAYLOAD; p.src0 = HDR; p.src_stride = FRAME;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> t;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0));
        int rc = bb_decode_frames(in, in_bytes, nullptr, nframes, &p, out, nframes * PAYLOAD * 4, nullptr);
        if (rc) { fprintf(stderr, "bb_decode_frames rc %d\n", rc); exit(1); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return (double)nframes * (FRAME + PAYLOAD * 16) / t[t.size() / 2] / 1e9;   // TB/s
}

int main(int argc, char **argv)
{
    size_t chunk_mib = argc > 1 ? strtoull(argv[1], 0, 10) : 2;
    size_t spacer_gib = argc > 2 ? strtoull(argv[2], 0, 10) : 24;
    const size_t nframes = 1 << 16;
    const size_t in_bytes = nframes * FRAME, out_bytes = nframes * PAYLOAD * 16;
    CK(hipSetDevice(0));
    if (bb_init()) { fprintf(stderr, "bb_init failed\n"); return 1; }
    void *in; CK(hipMalloc(&in, in_bytes + 256));
    hipLaunchKernelGGL(k_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)in, in_bytes / 4, 7u);
    CK(hipDeviceSynchronize());

    // A: plain allocation
    float *a; CK(hipMalloc(&a, out_bytes));
    printf("{\"case\": \"A hipMalloc\", \"TBps\": %.3f}\n", decode_rate(in, in_bytes, a, nframes));
    CK(hipFree(a));

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t chunk = chunk_mib << 20;
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t nchunk = (out_bytes + chunk - 1) / chunk;
    printf("{\"granularity\": %zu, \"chunk\": %zu, \"nchunk\": %zu}\n", gran, chunk, nchunk);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    for (int R : {1, 2, 4, 8}) {
        std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
        std::vector<void *> spacers;
        auto t0 = std::chrono::steady_clock::now();
        // group g owns chunks g, g + R, g + 2R, ...; groups are created one after
        // the other with a spacer allocation in between
        for (int g = 0; g < R; ++g) {
            for (size_t k = g; k < nchunk; k += R) CK(hipMemCreate(&h[k], chunk, &prop, 0));
            if (g + 1 < R) {
                void *s = nullptr;
                if (hipMalloc(&s, spacer_gib << 30) != hipSuccess) { fprintf(stderr, "spacer failed\n"); (void)hipGetLastError(); }
                else spacers.push_back(s);
            }
        }
        double create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, nchunk * chunk, 0, nullptr, 0));
        t0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < nchunk; ++k) CK(hipMemMap((char *)va + k * chunk, chunk, 0, h[k], 0));
        CK(hipMemSetAccess(va, nchunk * chunk, &acc, 1));
        double map_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double r1 = decode_rate(in, in_bytes, (float *)va, nframes);
        // spacers released: the chunks stay where they are
        for (void *s : spacers) CK(hipFree(s));
        double r2 = decode_rate(in, in_bytes, (float *)va, nframes);
        printf("{\"case\": \"VMM %d group(s), %zu GiB spacers\", \"TBps\": %.3f, \"TBps_after_spacers_freed\": %.3f, "
               "\"create_ms\": %.1f, \"map_ms\": %.1f}\n", R, R > 1 ? spacer_gib : 0, r1, r2, create_ms, map_ms);
        fflush(stdout);
        CK(hipMemUnmap(va, nchunk * chunk));
        for (size_t k = 0; k < nchunk; ++k) CK(hipMemRelease(h[k]));
        CK(hipMemAddressFree(va, nchunk * chunk));
    }
    return 0;
}
