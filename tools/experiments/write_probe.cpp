// write_probe: how fast can a kernel that ONLY writes go, and with which shape?  The decode
// kernels are bound by their stores (the headline launch with every source -1 -- stores only --
// runs at 0.90 of 8 TB/s).  Variants: 16-byte stores per lane and item (4 / 12 / 48 = 16 / 48 /
// 192 KiB per workgroup), plain / nontemporal, one item per workgroup or a persistent grid,
// items dealt over 1 / 8 stripes of the buffer.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/write_probe tools/experiments/write_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int N, bool NT>
__global__ __launch_bounds__(256)
void k_write(f4 *out, uint64_t nitems, uint64_t stripes, float seed)
{
    for (uint64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint64_t per = nitems / stripes;
        const uint64_t item = (it % stripes) * per + it / stripes;
        f4 *p = out + item * (uint64_t)(N * 256) + threadIdx.x;
        const f4 v = {seed, seed + 1.f, seed + 2.f, (float)threadIdx.x};
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            if (NT) __builtin_nontemporal_store(v, p + 256 * j);
            else p[256 * j] = v;
        }
    }
}

template <int N, bool NT>
static void run(f4 *d, uint64_t nbytes, int persistent_mult, uint64_t stripes)
{
    const uint64_t item_bytes = (uint64_t)N * 256 * 16;
    uint64_t nitems = nbytes / item_bytes;
    nitems -= nitems % stripes;
    const unsigned grid = persistent_mult ? 256u * persistent_mult : (unsigned)nitems;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_write<N, NT>), dim3(grid), dim3(256), 0, 0, d, nitems, stripes, 1.f);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_write<N, NT>), dim3(grid), dim3(256), 0, 0, d, nitems, stripes, (float)r);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double gbps = (double)nitems * item_bytes / ms / 1e6;
    printf("%3d KiB per workgroup  %-3s grid %-7s stripes %llu : %8.3f ms  %7.0f GB/s = %.3f of 8 TB/s\n", N * 4, NT ? "nt" : "",
           persistent_mult ? (persistent_mult == 8 ? "256x8" : "256x32") : "items", (unsigned long long)stripes, ms, gbps, gbps / 8000.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 64.0;
    const uint64_t nbytes = (uint64_t)(gib * 1073741824.0);
    f4 *d;
    CK(hipMalloc(&d, nbytes));
    CK(hipMemset(d, 0, nbytes));
    CK(hipDeviceSynchronize());
    for (uint64_t stripes : {1ull, 8ull}) {
        for (int mult : {0, 32}) {
            run<4, false>(d, nbytes, mult, stripes);
            run<4, true>(d, nbytes, mult, stripes);
            run<12, false>(d, nbytes, mult, stripes);
            run<12, true>(d, nbytes, mult, stripes);
            run<48, true>(d, nbytes, mult, stripes);
        }
    }
    return 0;
}
