// va_reuse_probe: when is it safe to map new memory at a virtual address that was
// unmapped before (csrc/bb_arena.inc, "VIRTUAL ADDRESSES ARE NEVER REUSED";
// tests/test_arena_gpu.py::test_a_thousand_grow_and_trim_cycles... failed when the
// arena freed a used-up range and the next reservation came back at the same address)?
// Cycle: reserve R GiB (optionally keeping the last Q reservations alive), create 1 GiB,
// map, fill with the cycle number, verify ALL of it with a second kernel, unmap, release,
// (optionally hipDeviceSynchronize), free the reservation (or keep it for Q cycles).
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/va_reuse_probe tools/experiments/va_reuse_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <deque>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_fill(uint32_t *p, size_t n, uint32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v ^ (uint32_t)i;
}
__global__ void k_check(const uint32_t *p, size_t n, uint32_t v, unsigned long long *bad)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (; i < n; i += stride) b += p[i] != (v ^ (uint32_t)i);
    if (b) atomicAdd(bad, b);
}

int main(int argc, char **argv)
{
    const int cycles = argc > 1 ? atoi(argv[1]) : 400;
    const int quarantine = argc > 2 ? atoi(argv[2]) : 0;     // reservations kept alive after use
    const int sync_after = argc > 3 ? atoi(argv[3]) : 0;     // 1: hipDeviceSynchronize after unmap + release
    const size_t GiB = 1ull << 30, chunk = 32u << 20, step = GiB;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long *bad = nullptr;
    CK(hipMalloc(&bad, 8));
    std::deque<void *> held;
    std::set<void *> seen;
    int reused = 0, failures = 0;
    unsigned long long bad_words = 0;
    for (int c = 0; c < cycles; ++c) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, step, chunk, nullptr, 0));
        if (!seen.insert(va).second) ++reused;
        std::vector<hipMemGenericAllocationHandle_t> h(step / chunk);
        for (auto &x : h) CK(hipMemCreate(&x, chunk, &prop, 0));
        for (size_t k = 0; k < h.size(); ++k) CK(hipMemMap((char *)va + k * chunk, chunk, 0, h[k], 0));
        CK(hipMemSetAccess(va, step, &acc, 1));
        CK(hipMemsetAsync(bad, 0, 8, nullptr));
        k_fill<<<4096, 256>>>((uint32_t *)va, step / 4, (uint32_t)c * 2654435761u);
        k_check<<<4096, 256>>>((const uint32_t *)va, step / 4, (uint32_t)c * 2654435761u, bad);
        unsigned long long b = 0;
        CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
        if (b) { ++failures; bad_words += b; if (failures <= 5) printf("cycle %d: %llu wrong words at %p\n", c, b, va); }
        for (size_t k = 0; k < h.size(); ++k) CK(hipMemUnmap((char *)va + k * chunk, chunk));
        for (auto x : h) CK(hipMemRelease(x));
        if (sync_after) CK(hipDeviceSynchronize());
        held.push_back(va);
        while ((int)held.size() > quarantine) { CK(hipMemAddressFree(held.front(), step)); held.pop_front(); }
    }
    printf("cycles %d quarantine %d sync %d: %d distinct addresses, %d reservations at an address seen before, "
           "%d cycles with wrong data (%llu words)\n", cycles, quarantine, sync_after, (int)seen.size(), reused, failures, bad_words);
    return 0;
}
