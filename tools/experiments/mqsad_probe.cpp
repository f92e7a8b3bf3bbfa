// What v_mqsad_u32_u8 computes on gfx950 and what it costs (round 5, locate sweep):
//   hipcc --offload-arch=gfx950 -O3 -o mqsad_probe mqsad_probe.cpp && ./mqsad_probe
// Prints which of two readings of the ISA text the hardware follows -- a zero byte of the
// 32-bit REFERENCE operand is left out of the sum (A), or a zero byte of the DATA (B) --
// and the issue rate of the instruction against v_xor/v_and pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void k_sem(const unsigned long long *in, unsigned pat, u4 *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u4 acc = {0u, 0u, 0u, 0u};
    out[i] = __builtin_amdgcn_mqsad_u32_u8(in[i], pat, acc);
}

template <int MODE>
__global__ void k_rate(unsigned long long seed, unsigned pat, unsigned *out, int iters)
{
    unsigned long long v = seed + threadIdx.x;
    u4 acc = {0u, 0u, 0u, 0u};
    u4 accs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) accs[k] = u4{0u, 0u, 0u, 0u};
    unsigned x = (unsigned)v, y = (unsigned)(v >> 32), r = 0xffffffffu;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            // 8 independent mqsads per trip
#pragma unroll
            for (int k = 0; k < 8; ++k)         // (eight independent accumulator chains)
                accs[k] = __builtin_amdgcn_mqsad_u32_u8(v + (unsigned long long)k * 0x0101010101ull, pat, accs[k]);
            v = v * 6364136223846793005ull + 1442695040888963407ull;
        } else {
            // the product's per-dword work: 3 alignbyte + 4 (xor, and) + mins
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned a = x + k, b = y + k;
                const unsigned m0 = (a ^ pat) & 0xffffff1fu;
                const unsigned m1 = (__builtin_amdgcn_alignbyte(b, a, 1) ^ pat) & 0xffffff1fu;
                const unsigned m2 = (__builtin_amdgcn_alignbyte(b, a, 2) ^ pat) & 0xffffff1fu;
                const unsigned m3 = (__builtin_amdgcn_alignbyte(b, a, 3) ^ pat) & 0xffffff1fu;
                const unsigned a01 = m0 < m1 ? m0 : m1, a23 = m2 < m3 ? m2 : m3;
                const unsigned m = a01 < a23 ? a01 : a23;
                r = r < m ? r : m;
            }
            x = x * 1664525u + 1013904223u; y = y * 22695477u + 1u;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc.x |= accs[k].x; acc.y |= accs[k].y; acc.z |= accs[k].z; acc.w |= accs[k].w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (acc.x | acc.y | acc.z | acc.w) + r;
}

static unsigned model(unsigned long long v, unsigned pat, int pos, int which)
{
    unsigned s = 0;
    for (int b = 0; b < 4; ++b) {
        const int d = (int)((v >> (8 * (pos + b))) & 0xff), p = (int)((pat >> (8 * b)) & 0xff);
        if (which == 0 && p == 0) continue;
        if (which == 1 && d == 0) continue;
        s += (unsigned)abs(d - p);
    }
    return s;
}

int main()
{
    const int n = 4096;
    std::vector<unsigned long long> h(n);
    srand(5);
    for (auto &x : h) {
        x = 0;
        for (int b = 0; b < 8; ++b) x |= (unsigned long long)((rand() % 4 == 0) ? 0 : (rand() & 0xff)) << (8 * b);
    }
    unsigned long long *d_in; u4 *d_out;
    hipMalloc(&d_in, n * 8); hipMalloc(&d_out, n * 16);
    hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
    for (unsigned pat : {0x0003ec00u, 0xabaddeedu, 0x000003ecu, 0xff00ff01u}) {
        k_sem<<<n / 256, 256>>>(d_in, pat, d_out, n);
        std::vector<u4> o(n);
        hipMemcpy(o.data(), d_out, n * 16, hipMemcpyDeviceToHost);
        int okA = 0, okB = 0, okN = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned got[4] = {o[i].x, o[i].y, o[i].z, o[i].w};
            bool a = true, b = true, c = true;
            for (int p = 0; p < 4; ++p) {
                a &= got[p] == model(h[i], pat, p, 0);
                b &= got[p] == model(h[i], pat, p, 1);
                c &= got[p] == model(h[i], pat, p, 2);
            }
            okA += a; okB += b; okN += c;
        }
        printf("pattern %08x: %d of %d as (A) zero REFERENCE bytes skipped, %d as (B) zero DATA bytes skipped, %d as plain SAD\n",
               pat, okA, n, okB, okN);
        if (pat == 0x0003ec00u)
            printf("  e.g. data %016llx -> %u %u %u %u\n", h[0], o[0].x, o[0].y, o[0].z, o[0].w);
    }
    unsigned *d_o; hipMalloc(&d_o, 1024 * 256 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 4096, blocks = 1024 * 8;
            hipEventRecord(e0);
            if (mode == 0) k_rate<0><<<blocks, 256>>>(12345, 0x0003ec00u, d_o, iters);
            else k_rate<1><<<blocks, 256>>>(12345, 0x0003ec00u, d_o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double dwords = (double)blocks * 256 * iters * 8;
            printf("%s: %.3f ms, %.1f G dword-starts (4 byte positions each) per second\n",
                   mode == 0 ? "v_mqsad_u32_u8" : "alignbyte/xor/and/min", ms, dwords / ms / 1e6);
        }
    }
    return 0;
}
