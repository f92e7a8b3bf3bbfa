/*
 * bbdecode_exp.h -- what only the EXPERIMENT build exports
 * (make -C baseband_amd/csrc EXPERIMENTS=1 -> baseband_amd/libbbdecode_exp.so,
 * loaded by tools/ with BB_EXPERIMENTS=1 in the environment): the kernel
 * variants rounds 1 and 2 measured against (csrc/k_exp.h, k_front.h, the
 * dispatcher in csrc/bb_exp.inc) and their knobs, a completion-time trace, and
 * the pinned-mapping copy helpers.  Nothing here is part of the product
 * library or of the drop-in boundary; the measurements are in docs/DESIGN_rounds1-3.md 3.2-3.3 and DESIGN.md 3.3.
 */
#ifndef BBDECODE_EXP_H
#define BBDECODE_EXP_H

#include <stddef.h>
#include <stdint.h>
#include "bbdecode_tune.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BB_TUNE_FLAT_VARIANT   0   /* 5 (default) = the product dispatch; 0 = plain kernel (workgroup per work item), 1 = byte loads (2-bit), 2 = persistent pipelined 4 waves x 8 tiles, 3 = 2 waves x 16 tiles, 4 = contiguous output cut in output space (k_decode_flat_span), 6-9 = explicit write front (k_decode_flat_front; slower), 10-12 = one pass with 2/4/8 stripes per wave (k_decode_flat_es; within +-4 % of 0 and 5), 14 = one float4 per thread and stripe (k_decode_flat_elem), 15 / 16 = the product dispatch with k_decode_flat_lds / k_decode_flat_lut forced for every sample width (A/B of the two byte-table kernels), 19 = k_decode_flat_lds with register-staged loads (round 3) instead of direct-to-LDS loads, 20 = k_decode_flat_lds with direct-to-LDS loads for every sample width (with BB_TUNE_FLAT8_LDS also 8-bit), 21 / 22 = the 2-bit kernel with 4 / 1 waves per workgroup instead of 2, 23 / 24 / 25 = its direct-to-LDS loads with the sc0 / nt / sc0+nt cache policy bits */
#define BB_TUNE_NT_STORES      1   /* 1 (default) = non-temporal stores; 0 = plain stores */
#define BB_TUNE_NT_LOADS       3   /* 1 = non-temporal input loads (no effect measured) */
#define BB_TUNE_TILES_PER_WAVE_8BIT 8 /* > 16: 8-bit contiguous data through the 32-tile instantiation of k_decode_flat_aln instead of the plain kernel */
#define BB_TUNE_LDS_PAD        9   /* bytes of unused dynamic LDS per workgroup of the flat kernels (caps workgroups per CU); 0 = none */
#define BB_TUNE_FRONT_GROUP   14   /* k_decode_flat_front: workgroups per group = width of the write front (default 2048) */
#define BB_TUNE_FRONT_STEPS   15   /* k_decode_flat_front: steps a group sweeps its region in (default 16) */
#define BB_TUNE_OUT_STRIPE_W  16   /* deal the frames of a contiguous-output launch over this many output regions (0 = off) ... */
#define BB_TUNE_OUT_STRIPE_S  17   /* ... that lie this many frame-slots apart: frame fs goes to slot (fs % W) * S + fs / W */
#define BB_TUNE_BYTE_LUT      21   /* 1 (default): contiguous 1-, 2- and 4-bit decode through k_decode_flat_lut; 0: k_decode_flat_aln (register level select) */
#define BB_TUNE_LUT_SMALL     25   /* 1: k_decode_flat_lut instantiated for at most 4 tiles per wave when the work items are that short (slower) */
#define BB_TUNE_FLAT8_LDS     29   /* 0 (default): the product dispatch (int8 through k_decode_flat_lds<8> with direct-to-LDS loads, VDIF 8-bit through k_decode_flat<8>); 2: k_decode_flat<8> for both; 1: contiguous 8-bit output of every coder through k_decode_flat_lds<8> (16-byte loads staged in LDS; BB_TUNE_LUT_TILES x 4 tiles per wave) instead of k_decode_flat<8>: -2..-5 % VDIF 8-bit, -1..+3 % int8 blocks (profiles/r03zd_exp_flat8*.log) */

#define BB_TUNE_M4_LDS        37   /* 1: Mark 4 units of 64-bit words through k_decode_mark4_lds (words staged in LDS by direct-to-LDS loads, no shuffles); default 0 = k_decode_mark4.  0.81 against 0.83 at 8 GiB in, +2.7 % at 2 GiB (profiles/r04s_exp_m4lds.log) */
#define BB_TUNE_COPY          35   /* k_copy_frames: (16-byte loads a lane has in flight: 2 / 4 / 8 / 16) | (non-temporal loads << 8); 0 = the product's 4 | 256 */
#define BB_TUNE_BURST         31   /* 1: contiguous 2-bit output through k_decode_flat_burst (a loader wave stages long work items in LDS with direct-to-LDS loads for 3 / 7 / 15 store waves); default 0.  Measured against k_decode_flat_lds (profiles/r04a_exp_burst.log): the same time to 0.2 % in the best configuration, 2-6 % slower with the clocked loader */
#define BB_TUNE_BURST_BYTES   32   /* ... LDS bytes per staging buffer (two per workgroup; 4096..79360, default 65536) */
#define BB_TUNE_BURST_PERIOD  33   /* ... loader waves issue their loads only at multiples of this many 10-ns wall-clock ticks (0 = at once) */
#define BB_TUNE_BURST_WAVES   34   /* ... store waves per workgroup: 3, 7 or 15 (default) */

/* When d_times is not NULL, the contiguous-output flat decode kernels store
 * the device wall clock (100 MHz ticks) at which each work item was completed
 * into d_times[item] (frames x work items per frame, plus one slot per
 * workgroup behind them, where the aligned kernel stores its start time). */
int bb_debug_trace(uint64_t *d_times);

/* bb_host_register pins a range of host memory where it lies -- e.g. a window
 * of a read-only file mapping -- so that bb_copy_to_device (hipMemcpyAsync on
 * `stream`) moves it without a host-side copy into a pinned buffer first
 * (tools/experiments/exp_hostregister.py; profiles/r02ax_exp_hostregister.log: pinning a
 * freshly mapped file costs more than the staging copy it would replace). */
int bb_host_register(const void *h_ptr, size_t nbytes);
int bb_host_unregister(const void *h_ptr);
int bb_copy_to_device(void *d_dst, const void *h_src, size_t nbytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BBDECODE_EXP_H */
