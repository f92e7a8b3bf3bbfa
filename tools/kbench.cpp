// kbench: standalone timing harness for libbbdecode kernels (developer tool).
// Times the flat decode on a synthetic fixed-stride VDIF-like file image
// (cfg2 geometry: 32-byte header + 8000-byte 2-bit payload) for each tuning
// combination, next to plain fill / copy kernels that bound what the HBM
// system delivers.  Usage: kbench [input MiB] [reps] [bps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include "bbdecode.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

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

template <bool NT>
__global__ void k_fill(f4 *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
    }
}

// Fill in "runs": a wave owns run r = wave + k * nwaves of RUN_KIB KiB and
// writes it 1 KiB per store instruction; threads per workgroup = blockDim.
template <int RUN_KIB>
__global__ void k_fill_runs(f4 *p, size_t nrun)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t r = wave; r < nrun; r += nwave) {
        f4 *q = p + r * (RUN_KIB * 64) + lane;
#pragma unroll 8
        for (int k = 0; k < RUN_KIB; ++k) __builtin_nontemporal_store(v, q + 64 * k);
    }
}

__global__ void k_copy(const f4 *s, f4 *d, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = s[i];
}

// Fill patterns that mimic the decode kernel's store order, to separate "what
// the write pattern costs" from "what the decode adds".
//  mode 0: one workgroup per 128000-byte region, waves interleaved per 4 KiB
//          (the decode kernel's order); mode 1: each wave owns a contiguous
//          quarter of the region; mode 2: as 0 plus the 8000-byte input read.
template <int MODE, bool NT>
__global__ __launch_bounds__(256)
void k_fill_regions(f4 *out, const uint32_t *in, size_t nregions)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t R4 = 8000;                 // float4 per region (128000 B)
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t r = blockIdx.x; r < nregions; r += gridDim.x) {
        f4 *o = out + r * R4;
        if (MODE == 1) {
            for (size_t i = wave * 2000 + lane; i < (size_t)(wave + 1) * 2000; i += 64) {
                if (NT) __builtin_nontemporal_store(v, &o[i]); else o[i] = v;
            }
        } else {
            uint32_t acc = 0;
            for (int t = wave; t < 32; t += 4) {            // 32 tiles of 256 float4
                if (MODE == 2) {
                    size_t dw = (size_t)t * 64 + lane;
                    if (dw < 2000) acc = in[r * 2008 + 8 + dw];
                    v.x = (float)(acc & 3);
                }
                for (int p = 0; p < 4; ++p) {
                    size_t i = (size_t)t * 256 + p * 64 + lane;
                    if (i < R4) { if (NT) __builtin_nontemporal_store(v, &o[i]); else o[i] = v; }
                }
            }
        }
    }
}

// "Elementwise" 2-bit decode: thread i produces float4 number i of the output
// (grid-stride over ALL float4 of the launch, exactly the store pattern of
// k_fill) from dword i / 4 of a headerless input: what does the dependent load
// cost when the write front is the fill's?  WPT float4 per thread (consecutive
// iterations of the grid-stride loop), loads issued before the stores.
template <int WPT>
__global__ void k_elem_decode(const uint32_t *in, f4 *out, size_t n4)
{
    const float lv[4] = {-3.316505f, -1.f, 1.f, 3.316505f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * WPT) {
        uint32_t w[WPT];
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const size_t i = i0 + k * stride;
            w[k] = i < n4 ? in[i >> 2] : 0u;
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const size_t i = i0 + k * stride;
            if (i >= n4) break;
            const uint32_t b = w[k] >> (8 * (i & 3));
            f4 v = {lv[b & 3], lv[(b >> 2) & 3], lv[(b >> 4) & 3], lv[(b >> 6) & 3]};
            __builtin_nontemporal_store(v, &out[i]);
        }
    }
}

// The same in ONE pass with the grid sized for it: a thread owns WPT float4.
// STRIPED: they lie n4 / WPT apart (WPT write fronts that advance with the
// dispatch order); otherwise they are consecutive 4 KiB blocks of one front
// (the workgroup writes WPT * 4 KiB contiguous).
template <int WPT, bool STRIPED, bool HDR = false>
__global__ void k_elem_decode1(const uint32_t *in, f4 *out, size_t n4)
{
    const float lv[4] = {-3.316505f, -1.f, 1.f, 3.316505f};
    const size_t per = (n4 + WPT - 1) / WPT;                // float4 per stripe
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w[WPT];
    size_t idx[WPT];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        idx[k] = STRIPED ? t + k * per
                         : ((size_t)blockIdx.x * WPT + k) * blockDim.x + threadIdx.x;
        const bool ok = STRIPED ? (t < per && idx[k] < n4) : idx[k] < n4;
        if (!ok) idx[k] = ~(size_t)0;
        if (HDR) {
            // real frames: 2000 payload dwords behind an 8-dword header (8032-byte
            // frames); float4 i of the output belongs to frame i / 8000
            const size_t f = idx[k] / 8000, r = idx[k] - f * 8000;
            w[k] = ok ? in[f * 2008 + 8 + (r >> 2)] : 0u;
        } else {
            w[k] = ok ? in[idx[k] >> 2] : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        if (idx[k] == ~(size_t)0) continue;
        const uint32_t b = w[k] >> (8 * (idx[k] & 3));
        f4 v = {lv[b & 3], lv[(b >> 2) & 3], lv[(b >> 4) & 3], lv[(b >> 6) & 3]};
        __builtin_nontemporal_store(v, &out[idx[k]]);
    }
}

// Store flavours (MI355X_MICROARCH.md, "stores of each flavour"): plain / nt keep
// the line in the XCD's L2, sc1 / sc0 sc1 drop it after the write.  Same
// grid-stride fill, the store written in inline asm.  MODE 0 plain, 1 nt,
// 2 sc1, 3 sc0 sc1, 4 sc1 nt.
template <int MODE>
__global__ void k_fill_flavour(f4 *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n; i += stride) {
        f4 *q = p + i;
        if (MODE == 0)      asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(q), "v"(v) : "memory");
        else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q), "v"(v) : "memory");
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
        else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
        else                asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(q), "v"(v) : "memory");
    }
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv)
{
    size_t in_mib = argc > 1 ? strtoull(argv[1], 0, 10) : 2048;
    int reps = argc > 2 ? atoi(argv[2]) : 5;
    int bps = argc > 3 ? atoi(argv[3]) : 2;
    const size_t frame = 8032, payload = 8000, hdr = 32;
    size_t nframes = (in_mib << 20) / frame;
    size_t in_bytes = nframes * frame;
    size_t E = payload * 8 / bps;
    size_t out_elems = nframes * E;
    printf("kbench: %zu frames, in %.3f GiB, out %.3f GiB, bps %d\n", nframes,
           in_bytes / 1073741824.0, out_elems * 4 / 1073741824.0, bps);

    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, %.1f GiB\n", prop.name, prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0);

    uint8_t *d_in; float *d_out;
    CK(hipMalloc(&d_in, in_bytes + 256));
    CK(hipMalloc(&d_out, out_elems * 4));
    hipLaunchKernelGGL(k_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)d_in, in_bytes / 4, 12345u);
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg_bytes = (double)in_bytes + (double)out_elems * 4;

    // the fill bound against the grid size (grid-stride float4 fill, nt stores)
    for (unsigned gridsz : {2048u, 8192u, 32768u, 131072u, 524288u, 2097152u}) {
        std::vector<double> t;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_fill<true>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (r) t.push_back(time_ms(e0, e1));
        }
        std::sort(t.begin(), t.end());
        printf("fill nt=1 grid=%u: median %.3f ms  %.1f GB/s\n", gridsz, t[t.size() / 2],
               out_elems * 4 / t[t.size() / 2] / 1e6);
    }
    if (getenv("KB_STORE")) {
        for (unsigned gridsz : {32768u, 2097152u}) {
            for (int mode = 0; mode < 5; ++mode) {
                std::vector<double> t;
                for (int r = 0; r < reps + 1; ++r) {
                    CK(hipEventRecord(e0));
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(k_fill_flavour<0>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4); break;
                        case 1: hipLaunchKernelGGL(k_fill_flavour<1>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4); break;
                        case 2: hipLaunchKernelGGL(k_fill_flavour<2>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4); break;
                        case 3: hipLaunchKernelGGL(k_fill_flavour<3>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4); break;
                        default: hipLaunchKernelGGL(k_fill_flavour<4>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4); break;
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    if (r) t.push_back(time_ms(e0, e1));
                }
                std::sort(t.begin(), t.end());
                static const char *names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt"};
                printf("fill store=%s grid=%u: %.3f ms  %.1f GB/s\n", names[mode], gridsz, t[t.size() / 2],
                       out_elems * 4 / t[t.size() / 2] / 1e6);
            }
        }
        return 0;
    }
    if (getenv("KB_ELEM")) {
        // elementwise decode against the fill, same grids
        const size_t n4 = std::min<size_t>((in_bytes / 4) * 4, out_elems / 4);   // float4 count = 4 per input dword, inside the output buffer
        for (unsigned gridsz : {8192u, 32768u, 131072u, 524288u, 2097152u, 0u}) {
            for (int wpt : {1, 4}) {
                size_t need = (n4 + 255) / 256;
                unsigned g = gridsz ? gridsz : (unsigned)std::min<size_t>(need, 0x7fffffff);
                if (gridsz && (size_t)gridsz > need) continue;
                std::vector<double> t;
                for (int r = 0; r < reps + 1; ++r) {
                    CK(hipEventRecord(e0));
                    if (wpt == 1) hipLaunchKernelGGL(k_elem_decode<1>, dim3(g), dim3(256), 0, 0, (const uint32_t *)d_in, (f4 *)d_out, n4);
                    else          hipLaunchKernelGGL(k_elem_decode<4>, dim3(g), dim3(256), 0, 0, (const uint32_t *)d_in, (f4 *)d_out, n4);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    if (r) t.push_back(time_ms(e0, e1));
                }
                std::sort(t.begin(), t.end());
                printf("elem_decode grid=%u%s wpt=%d: %.3f ms  %.1f GB/s (in+out)\n", g, gridsz ? "" : " (one-shot)", wpt,
                       t[t.size() / 2], (n4 * 16.0 + n4) / t[t.size() / 2] / 1e6);
            }
        }
#define KB_ONE(W, S) do { \
            const size_t need = ((n4 + (W) - 1) / (W) + 255) / 256; \
            std::vector<double> t; \
            for (int r = 0; r < reps + 1; ++r) { \
                CK(hipEventRecord(e0)); \
                hipLaunchKernelGGL((k_elem_decode1<W, S>), dim3((unsigned)need), dim3(256), 0, 0, (const uint32_t *)d_in, (f4 *)d_out, n4); \
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
                if (r) t.push_back(time_ms(e0, e1)); \
            } \
            std::sort(t.begin(), t.end()); \
            printf("elem_decode one pass wpt=%d %s: %.3f ms  %.1f GB/s (in+out)\n", W, (S) ? "striped" : "contiguous", \
                   t[t.size() / 2], (n4 * 16.0 + n4) / t[t.size() / 2] / 1e6); } while (0)
#define KB_ONE_H(W) do { \
            const size_t n4h = std::min<size_t>(nframes * 8000, out_elems / 4); \
            const size_t need = ((n4h + (W) - 1) / (W) + 255) / 256; \
            std::vector<double> t; \
            for (int r = 0; r < reps + 1; ++r) { \
                CK(hipEventRecord(e0)); \
                hipLaunchKernelGGL((k_elem_decode1<W, true, true>), dim3((unsigned)need), dim3(256), 0, 0, (const uint32_t *)d_in, (f4 *)d_out, n4h); \
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
                if (r) t.push_back(time_ms(e0, e1)); \
            } \
            std::sort(t.begin(), t.end()); \
            printf("elem_decode one pass wpt=%d striped, 8032-byte frames (header skipped, division per piece): %.3f ms  %.1f GB/s (in+out)\n", W, \
                   t[t.size() / 2], (n4h * 16.0 + nframes * 8032.0) / t[t.size() / 2] / 1e6); } while (0)
        KB_ONE_H(8); KB_ONE_H(16);
        KB_ONE(2, true); KB_ONE(4, true); KB_ONE(8, true); KB_ONE(16, true); KB_ONE(32, true);
        KB_ONE(2, false); KB_ONE(4, false); KB_ONE(8, false); KB_ONE(16, false);
        for (unsigned gridsz : {32768u, 2097152u}) {
            std::vector<double> t;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_fill<true>, dim3(gridsz), dim3(256), 0, 0, (f4 *)d_out, n4);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                if (r) t.push_back(time_ms(e0, e1));
            }
            std::sort(t.begin(), t.end());
            printf("fill same range grid=%u: %.3f ms  %.1f GB/s\n", gridsz, t[t.size() / 2], n4 * 16.0 / t[t.size() / 2] / 1e6);
        }
        return 0;
    }
    // fill by runs: run length x threads per workgroup x grid
    if (getenv("KB_RUNS")) {
        for (int run_kib : {1, 4, 16, 32, 64}) {
            for (int threads : {64, 128, 256}) {
                for (unsigned gridsz : {4096u, 32768u, 131072u, 524288u, 2097152u, 0u}) {
                    const size_t nrun = out_elems * 4 / ((size_t)run_kib * 1024);
                    size_t need = (nrun * 64 + threads - 1) / threads;
                    unsigned g = gridsz ? gridsz : (unsigned)std::min<size_t>(need, 0x7fffffff);
                    if (gridsz && (size_t)gridsz > need) continue;
                    std::vector<double> t;
                    for (int r = 0; r < 4; ++r) {
                        CK(hipEventRecord(e0));
                        switch (run_kib) {
                            case 1:  hipLaunchKernelGGL(k_fill_runs<1>, dim3(g), dim3(threads), 0, 0, (f4 *)d_out, nrun); break;
                            case 4:  hipLaunchKernelGGL(k_fill_runs<4>, dim3(g), dim3(threads), 0, 0, (f4 *)d_out, nrun); break;
                            case 16: hipLaunchKernelGGL(k_fill_runs<16>, dim3(g), dim3(threads), 0, 0, (f4 *)d_out, nrun); break;
                            case 32: hipLaunchKernelGGL(k_fill_runs<32>, dim3(g), dim3(threads), 0, 0, (f4 *)d_out, nrun); break;
                            default: hipLaunchKernelGGL(k_fill_runs<64>, dim3(g), dim3(threads), 0, 0, (f4 *)d_out, nrun); break;
                        }
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        if (r) t.push_back(time_ms(e0, e1));
                    }
                    std::sort(t.begin(), t.end());
                    printf("fill_runs run=%dKiB threads=%d grid=%u%s: %.3f ms  %.1f GB/s\n", run_kib, threads, g,
                           gridsz ? "" : " (one run per wave)", t[t.size() / 2], out_elems * 4 / t[t.size() / 2] / 1e6);
                }
            }
        }
        return 0;
    }
    // bounds: fill and copy
    for (int nt = 0; nt < 2; ++nt) {
        std::vector<double> t;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            if (nt) hipLaunchKernelGGL(k_fill<true>, dim3(256 * 8), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4);
            else    hipLaunchKernelGGL(k_fill<false>, dim3(256 * 8), dim3(256), 0, 0, (f4 *)d_out, out_elems / 4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (r) t.push_back(time_ms(e0, e1));
        }
        std::sort(t.begin(), t.end());
        printf("fill nt=%d: median %.3f ms  %.1f GB/s\n", nt, t[t.size() / 2], out_elems * 4 / t[t.size() / 2] / 1e6);
    }
    {
        std::vector<double> t;
        size_t n = out_elems / 8;   // copy first half into second half
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const f4 *)d_out, (f4 *)d_out + n, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (r) t.push_back(time_ms(e0, e1));
        }
        std::sort(t.begin(), t.end());
        printf("copy: median %.3f ms  %.1f GB/s (read+write)\n", t[t.size() / 2], n * 32.0 / t[t.size() / 2] / 1e6);
    }

    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> t;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            dim3 g((unsigned)nframes), b(256);
            if (mode == 0) hipLaunchKernelGGL((k_fill_regions<0, true>), g, b, 0, 0, (f4 *)d_out, (const uint32_t *)d_in, nframes);
            if (mode == 1) hipLaunchKernelGGL((k_fill_regions<1, true>), g, b, 0, 0, (f4 *)d_out, (const uint32_t *)d_in, nframes);
            if (mode == 2) hipLaunchKernelGGL((k_fill_regions<2, true>), g, b, 0, 0, (f4 *)d_out, (const uint32_t *)d_in, nframes);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (r) t.push_back(time_ms(e0, e1));
        }
        std::sort(t.begin(), t.end());
        double bytes = (double)out_elems * 4 + (mode == 2 ? (double)in_bytes : 0.0);
        printf("fill_regions mode=%d nt=1: median %.3f ms  %.1f GB/s\n", mode, t[t.size() / 2], bytes / t[t.size() / 2] / 1e6);
    }

    bb_decode_params p = {};
    p.coder = BB_CODER_VDIF; p.bps = bps; p.chunk = 1; p.nslot = 1;
    p.payload_nbytes = payload; p.src0 = hdr; p.src_stride = frame;
    p.complex_data = 0; p.fill_re = 0.f; p.fill_im = 0.f;

    const int blocks_opts[] = {0, 2048, 8192, 16384};
    for (int variant = 0; variant < 3; ++variant)
    for (int nt = 0; nt < 2; ++nt)
    for (int bi = 0; bi < 4; ++bi) {
        if (variant == 1 && bps != 2) continue;
        if (variant != 2 && (nt == 0 || bi > 1)) continue;      // keep the sweep short
        bb_tune(BB_TUNE_FLAT_VARIANT, variant);
        bb_tune(BB_TUNE_NT_STORES, nt);
        bb_tune(BB_TUNE_BLOCKS, blocks_opts[bi]);
        std::vector<double> t;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            int rc = bb_decode_frames(d_in, in_bytes, nullptr, nframes, &p, d_out, out_elems, nullptr);
            if (rc) { fprintf(stderr, "bb_decode_frames rc=%d (%s) hip=%d\n", rc, bb_strerror(rc), bb_last_hip_error()); return 1; }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (r) t.push_back(time_ms(e0, e1));
        }
        std::sort(t.begin(), t.end());
        double med = t[t.size() / 2];
        printf("decode variant=%d nt=%d blocks=%d: median %.3f ms (min %.3f)  %.1f GB/s alg  %.1f Msamples/s  frac8TB=%.3f\n",
               variant, nt, blocks_opts[bi], med, t[0], alg_bytes / med / 1e6, out_elems / med / 1e3,
               alg_bytes / med / 1e6 / 8000.0);
    }

    for (int ntl = 0; ntl < 6; ++ntl) {
        const int var_[6] = {2, 3, 3, 3, 2, 3}; const int blk_[6] = {0, 0, 8192, 32768, 8192, 0};
        bb_tune(BB_TUNE_FLAT_VARIANT, var_[ntl]); bb_tune(BB_TUNE_NT_STORES, 1);
        bb_tune(BB_TUNE_BLOCKS, blk_[ntl]);
        bb_tune(BB_TUNE_NT_LOADS, 0);
        std::vector<double> t;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            bb_decode_frames(d_in, in_bytes, nullptr, nframes, &p, d_out, out_elems, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (r) t.push_back(time_ms(e0, e1));
        }
        std::sort(t.begin(), t.end());
        printf("decode pipe exp=%d (0: 4 waves; 1,2: 2 waves/item default,16384 blocks; 3,4,5: 1 wave/item default,32768,8192 blocks): median %.3f ms  %.1f GB/s alg  frac8TB=%.3f\n", ntl,
               t[t.size() / 2], alg_bytes / t[t.size() / 2] / 1e6, alg_bytes / t[t.size() / 2] / 1e6 / 8000.0);
    }
    bb_tune(BB_TUNE_NT_LOADS, 0);

    // spot check (2-bit VDIF only): first and last frame against a host LUT
    if (bps == 2) {
        bb_tune(BB_TUNE_FLAT_VARIANT, 0); bb_tune(BB_TUNE_NT_STORES, 0); bb_tune(BB_TUNE_BLOCKS, 0);
        bb_decode_frames(d_in, in_bytes, nullptr, nframes, &p, d_out, out_elems, nullptr);
        CK(hipDeviceSynchronize());
        float lv[4]; bb_get_levels(BB_CODER_VDIF, 2, lv, 4);
        size_t bad = 0;
        const size_t fr[3] = {0, nframes / 2, nframes - 1};
        std::vector<uint8_t> hb(payload); std::vector<float> ho(E);
        for (int k = 0; k < 3; ++k) {
            size_t f = fr[k];
            CK(hipMemcpy(hb.data(), d_in + f * frame + hdr, payload, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ho.data(), d_out + f * E, E * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < payload; ++i)
                for (int j = 0; j < 4; ++j)
                    if (ho[i * 4 + j] != lv[(hb[i] >> (2 * j)) & 3]) ++bad;
        }
        printf("spot check: %zu mismatches\n", bad);
    }
    return 0;
}
