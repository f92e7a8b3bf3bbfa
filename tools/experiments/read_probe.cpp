// read_probe: how fast can a kernel that ONLY reads go on this device, and with which shape?
// The encoders and the locate sweep are bound by reads; the one read-only figure on record is
// the locate sweep with its compares cut out (0.795 of 8 TB/s).  Variants: 16-byte loads in
// flight per lane (1 / 2 / 4 / 8), plain / nontemporal, one item per workgroup (grid = items)
// or a persistent grid, items dealt over 1 or 8 stripes of the buffer.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/read_probe tools/experiments/read_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// an item = FLY x 256 lanes x 16 bytes, contiguous; lane l of a workgroup takes the l-th 16 bytes of
// each of the item's FLY runs of 4 KiB
template <int FLY, bool NT>
__global__ __launch_bounds__(256)
void k_read(const u4 *in, uint64_t nitems, uint64_t stripes, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        // deal over stripes: item i -> stripe i % S, place i / S inside it
        const uint64_t per = nitems / stripes;
        const uint64_t item = (it % stripes) * per + it / stripes;
        const u4 *p = in + item * (uint64_t)(FLY * 256) + threadIdx.x;
        u4 v[FLY];
#pragma unroll
        for (int j = 0; j < FLY; ++j)
            v[j] = NT ? __builtin_nontemporal_load(p + 256 * j) : p[256 * j];
#pragma unroll
        for (int j = 0; j < FLY; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;          // (never: keeps the loads)
}

template <int FLY, bool NT>
static void run(const u4 *d, uint64_t nbytes, uint32_t *sink, int persistent_mult, uint64_t stripes)
{
    const uint64_t item_bytes = (uint64_t)FLY * 256 * 16;
    uint64_t nitems = nbytes / item_bytes;
    nitems -= nitems % stripes;
    const unsigned grid = persistent_mult ? 256u * persistent_mult : (unsigned)(nitems > 0x7fffffffull ? 0x7fffffff : nitems);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_read<FLY, NT>), dim3(grid), dim3(256), 0, 0, d, nitems, stripes, sink);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<FLY, NT>), dim3(grid), dim3(256), 0, 0, d, nitems, stripes, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double gbps = (double)nitems * item_bytes / ms / 1e6;
    printf("fly %d  %-3s grid %-10s stripes %llu : %8.3f ms  %7.0f GB/s = %.3f of 8 TB/s\n", FLY, NT ? "nt" : "",
           persistent_mult ? (persistent_mult == 4 ? "256x4" : persistent_mult == 8 ? "256x8" : persistent_mult == 16 ? "256x16" : "256x32") : "items",
           (unsigned long long)stripes, ms, gbps, gbps / 8000.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const uint64_t nbytes = (uint64_t)(gib * 1073741824.0);
    u4 *d; uint32_t *sink;
    CK(hipMalloc(&d, nbytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 0x5a, nbytes));
    CK(hipDeviceSynchronize());
    for (uint64_t stripes : {1ull, 8ull}) {
        for (int mult : {0, 8, 32}) {
            run<1, false>(d, nbytes, sink, mult, stripes);
            run<2, false>(d, nbytes, sink, mult, stripes);
            run<4, false>(d, nbytes, sink, mult, stripes);
            run<8, false>(d, nbytes, sink, mult, stripes);
            run<4, true>(d, nbytes, sink, mult, stripes);
            run<8, true>(d, nbytes, sink, mult, stripes);
            run<16, true>(d, nbytes, sink, mult, stripes);
        }
    }
    return 0;
}
