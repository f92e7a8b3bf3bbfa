// graph_probe: what would a HIP graph save on the launch-bound inner loop of a
// small read()?  The loop is four launches -- bb_vdif_scan, bb_build_index,
// bb_verify_records, bb_decode_frames -- for a couple of frames.  Host time per
// iteration of (a) the four library calls, (b) one hipGraphLaunch of the same
// sequence captured once (the upper bound: no parameter updates; a real read
// changes offsets and pointers every time and would have to update the nodes).
// Usage: graph_probe [frames per read, default 2]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <chrono>
#include "bbdecode.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define BB(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "bb error %d at %s:%d\n", r_, __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char **argv)
{
    const size_t nf = argc > 1 ? (size_t)atol(argv[1]) : 2;
    const size_t frame = 8032, payload = 8000;
    BB(bb_init());
    uint8_t *buf; bb_frame_rec *recs; int64_t *src; float *out; uint32_t *nbad;
    CK(hipMalloc(&buf, nf * frame + 256)); CK(hipMemset(buf, 0, nf * frame + 256));
    CK(hipMalloc(&recs, nf * sizeof(bb_frame_rec))); CK(hipMalloc(&src, nf * 8));
    CK(hipMalloc(&out, nf * payload * 16)); CK(hipMalloc(&nbad, 4)); CK(hipMemset(nbad, 0, 4));
    bb_vdif_scan_params sp; memset(&sp, 0, sizeof(sp));
    sp.frame_nbytes = (uint32_t)frame; sp.header_nbytes = 32; sp.frame_rate = 1000;
    bb_decode_params dp; memset(&dp, 0, sizeof(dp));
    dp.coder = BB_CODER_VDIF; dp.bps = 2; dp.chunk = 1; dp.nslot = 1; dp.payload_nbytes = payload;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto seq = [&]() {
        BB(bb_vdif_scan(buf, nf * frame, &sp, recs, nf, st));
        BB(bb_build_index(recs, nf, nullptr, 1, src, nf, st));
        BB(bb_verify_records(recs, nf, 0, 1, nf, nbad, st));
        BB(bb_decode_frames(buf, nf * frame, src, nf, &dp, out, nf * payload * 4, st));
    };
    for (int i = 0; i < 50; ++i) seq();
    CK(hipStreamSynchronize(st));
    const int N = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) seq();
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    seq();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    auto t3 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) CK(hipGraphLaunch(ge, st));
    auto t4 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t5 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("{\"frames_per_read\": %zu, \"four_calls_host_us\": %.2f, \"four_calls_done_us\": %.2f, "
           "\"graph_launch_host_us\": %.2f, \"graph_launch_done_us\": %.2f}\n",
           nf, us(t0, t1) / N, us(t0, t2) / N, us(t3, t4) / N, us(t3, t5) / N);
    return 0;
}
