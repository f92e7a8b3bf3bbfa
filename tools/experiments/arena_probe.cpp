// arena_probe: which physical chunks of HBM should a decode OUTPUT be made of?
//
// Round 2 found that a decode-shaped store stream inside one physical "region"
// of HBM runs at 5.4-5.65 TB/s and at 6.3-6.9 when it covers different regions
// at the same time, and that an output mapped from 2 MiB VMM
// chunks was 12-18 % faster than a plain hipMalloc (profiles/r02j_vmm_probe.log).
// This probe takes (nearly) ALL free HBM as 2 MiB hipMemCreate chunks, in
// creation order ("groups" of GROUP_CHUNKS consecutive chunks), and measures
// the cfg2 decode (8032-byte frames, 2-bit) into outputs mapped from chosen
// chunks:
//   pairs   T(ref, j): 2 GiB output alternating chunks of group ref and group j
//   layouts 2^15 / 2^16 / 2^18-frame outputs made of: one run of consecutive
//           groups (= what hipMalloc gives); chunks dealt round robin over
//           K groups spread evenly over all of memory, in runs of 1 or 16 chunks
// Build: make -C baseband_amd/csrc arena_probe.   Usage: arena_probe [mode ...]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
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
static const size_t CHUNK = 2u << 20;
static size_t GROUP_CHUNKS = 1024;           // 2 GiB

typedef std::chrono::steady_clock clk;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

static hipEvent_t e0, e1;

static double decode_rate(const void *in, size_t in_bytes, float *out, size_t nframes, int reps = 4)
{
    bb_decode_params p = {};
    p.coder = BB_CODER_VDIF; p.bps = 2; p.chunk = 1; p.nslot = 1;
    p.payload_nbytes = PAYLOAD; p.src0 = HDR; p.src_stride = FRAME;
    std::vector<double> t;
    for (int r = 0; r <= reps; ++r) {
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

static hipMemAccessDesc acc;

static void map_chunks(char *va, const std::vector<hipMemGenericAllocationHandle_t> &h)
{
    for (size_t k = 0; k < h.size(); ++k) CK(hipMemMap(va + k * CHUNK, CHUNK, 0, h[k], 0));
    CK(hipMemSetAccess(va, h.size() * CHUNK, &acc, 1));
}

int main(int argc, char **argv)
{
    bool do_pairs = false, do_layouts = false, do_base = false;
    size_t keep_gib = 6;                      // HBM left alone (input, runtime)
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "pairs")) do_pairs = true;
        else if (!strcmp(argv[i], "layouts")) do_layouts = true;
        else if (!strcmp(argv[i], "base")) do_base = true;
        else if (!strncmp(argv[i], "keep=", 5)) keep_gib = strtoull(argv[i] + 5, 0, 10);
        else if (!strncmp(argv[i], "group=", 6)) GROUP_CHUNKS = strtoull(argv[i] + 6, 0, 10);
    }
    CK(hipSetDevice(0));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (bb_init()) { fprintf(stderr, "bb_init failed\n"); return 1; }
    const size_t max_frames = 1 << 18;
    const size_t in_frames = 1 << 16;                    // the input is reused by every launch size
    const size_t in_bytes = in_frames * FRAME;
    void *in; CK(hipMalloc(&in, max_frames * FRAME + 256));
    hipLaunchKernelGGL(k_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)in, max_frames * FRAME / 4, 7u);
    CK(hipDeviceSynchronize());
    (void)in_bytes;

    if (do_base) {
        // what plain allocations give, a fresh one per measurement
        for (size_t lf : {15, 16, 18}) {
            const size_t nf = (size_t)1 << lf;
            for (int r = 0; r < 4; ++r) {
                float *a; CK(hipMalloc(&a, nf * PAYLOAD * 16));
                printf("{\"case\": \"hipMalloc\", \"log2_frames\": %zu, \"draw\": %d, \"TBps\": %.3f}\n", lf, r,
                       decode_rate(in, nf * FRAME, a, nf));
                fflush(stdout);
                CK(hipFree(a));
            }
        }
    }

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    size_t nchunk = (free_b - (keep_gib << 30)) / CHUNK;
    nchunk = nchunk / GROUP_CHUNKS * GROUP_CHUNKS;
    const size_t NG = nchunk / GROUP_CHUNKS;
    std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
    auto t0 = clk::now();
    size_t made = 0;
    for (; made < nchunk; ++made)
        if (hipMemCreate(&h[made], CHUNK, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    const double create_ms = ms_since(t0);
    printf("{\"granularity\": %zu, \"free_GiB\": %.1f, \"total_GiB\": %.1f, \"chunks\": %zu, \"made\": %zu, \"groups\": %zu, "
           "\"group_GiB\": %.2f, \"create_ms\": %.1f}\n", gran, free_b / 1073741824.0, total_b / 1073741824.0,
           nchunk, made, NG, GROUP_CHUNKS * CHUNK / 1073741824.0, create_ms);
    fflush(stdout);
    if (made < nchunk) { fprintf(stderr, "hipMemCreate stopped at %zu\n", made); return 1; }

    const size_t va_bytes = ((max_frames * PAYLOAD * 16 + CHUNK - 1) / CHUNK + 16) * CHUNK;
    char *va = nullptr;
    CK(hipMemAddressReserve((void **)&va, va_bytes, 0, nullptr, 0));

    if (do_pairs) {
        // T(ref, j): 2^14 frames (2.1 GB) into chunks alternating between the two groups
        const size_t nf = 1 << 14;
        const size_t need = (nf * PAYLOAD * 16 + CHUNK - 1) / CHUNK;      // 1000 chunks
        if (need > GROUP_CHUNKS) { fprintf(stderr, "group too small\n"); return 1; }
        const size_t refs[] = {0, NG / 4, NG / 2, 3 * NG / 4, NG - 1};
        for (size_t ref : refs) {
            printf("{\"pairs_ref\": %zu, \"TBps\": [", ref);
            for (size_t j = 0; j < NG; ++j) {
                std::vector<hipMemGenericAllocationHandle_t> l;
                for (size_t k = 0; k < need; ++k) {
                    // same group: its first `need` chunks in order
                    if (j == ref) l.push_back(h[ref * GROUP_CHUNKS + k % GROUP_CHUNKS]);
                    else l.push_back(h[((k & 1) ? j : ref) * GROUP_CHUNKS + k / 2]);
                }
                map_chunks(va, l);
                const double r = decode_rate(in, nf * FRAME, (float *)va, nf, 3);
                CK(hipMemUnmap(va, l.size() * CHUNK));
                printf("%s%.2f", j ? ", " : "", r);
                fflush(stdout);
            }
            printf("]}\n");
            fflush(stdout);
        }
    }

    if (do_layouts) {
        for (size_t lf : {15, 16, 18}) {
            const size_t nf = (size_t)1 << lf;
            const size_t need = (nf * PAYLOAD * 16 + CHUNK - 1) / CHUNK;
            // (a) consecutive chunks starting at several groups
            for (size_t g0 = 0; g0 + (need + GROUP_CHUNKS - 1) / GROUP_CHUNKS <= NG; g0 += NG / 8 ? NG / 8 : 1) {
                std::vector<hipMemGenericAllocationHandle_t> l(h.begin() + g0 * GROUP_CHUNKS, h.begin() + g0 * GROUP_CHUNKS + need);
                t0 = clk::now();
                map_chunks(va, l);
                const double map_ms = ms_since(t0);
                const double r = decode_rate(in, nf * FRAME, (float *)va, nf);
                t0 = clk::now();
                CK(hipMemUnmap(va, l.size() * CHUNK));
                printf("{\"layout\": \"consecutive\", \"log2_frames\": %zu, \"first_group\": %zu, \"TBps\": %.3f, \"map_ms\": %.1f, \"unmap_ms\": %.1f}\n",
                       lf, g0, r, map_ms, ms_since(t0));
                fflush(stdout);
            }
            // (b) round robin over K groups spread evenly over memory, runs of RUN chunks
            for (size_t K : {2, 3, 4, 6, 8, 16, 32, 64, 0}) {
                const size_t KK = K ? K : NG;
                if (KK > NG) continue;
                for (size_t RUN : {1, 16}) {
                    for (size_t shift : {0, 1}) {
                        // group of the i-th run: spread evenly, `shift` moves the comb by half a tooth
                        // (a tooth is a run of consecutive chunks starting at its group)
                        std::vector<size_t> used(KK, 0);
                        std::vector<hipMemGenericAllocationHandle_t> l;
                        bool fits = true;
                        for (size_t k = 0; k < need && fits; ++k) {
                            const size_t slot = (k / RUN) % KK;
                            const size_t g = (slot * NG) / KK + (shift ? NG / KK / 2 : 0);
                            const size_t c = g * GROUP_CHUNKS + used[slot]++;
                            if (c >= nchunk) fits = false; else l.push_back(h[c]);
                        }
                        if (!fits) continue;
                        map_chunks(va, l);
                        const double r = decode_rate(in, nf * FRAME, (float *)va, nf);
                        CK(hipMemUnmap(va, l.size() * CHUNK));
                        printf("{\"layout\": \"round_robin\", \"log2_frames\": %zu, \"groups\": %zu, \"run_chunks\": %zu, \"shift\": %zu, \"TBps\": %.3f}\n",
                               lf, KK, RUN, shift, r);
                        fflush(stdout);
                    }
                }
            }
        }
    }
    return 0;
}
