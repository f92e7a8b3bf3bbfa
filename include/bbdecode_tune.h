/*
 * bbdecode_tune.h -- geometry knobs of libbbdecode.so.
 *
 * Not part of the drop-in boundary (include/bbdecode.h is): these exist so that
 * tests can force every code path of a kernel (a small grid makes workgroups
 * loop; the striped work order on and off; both forms of a dispatch choice) and
 * so that benchmarks can re-measure a default.  Results never depend on a knob.
 * The experiment build (make -C baseband_amd/csrc EXPERIMENTS=1 ->
 * libbbdecode_exp.so) adds the knobs of include/bbdecode_exp.h.
 */
#ifndef BBDECODE_TUNE_H
#define BBDECODE_TUNE_H

#ifdef __cplusplus
extern "C" {
#endif

#define BB_TUNE_BLOCKS         2   /* 0 = default grid; >0 = upper bound of workgroups per launch */
#define BB_TUNE_TILE_ELEMS     4   /* elements per tile of k_decode_i8_tiled / _stage (default 8192) */
#define BB_TUNE_ENCODE_DIRECT  5   /* 1 = 2-bit encoders evaluate the reference clip/add/floor_divide arithmetic per sample instead of comparing with the three thresholds derived from it (check mode) */
#define BB_TUNE_GATHER_BYTES   6   /* payload bytes of all thread slots staged in LDS per work item of k_decode_gather; 8192 (default) = automatic: 16384, or 4096 for 1-bit data */
#define BB_TUNE_TILES_PER_WAVE 7   /* upper bound of 256-byte tiles per work item of k_decode_rows_pipe (1..8, default 8) */
#define BB_TUNE_TILED_STAGE   10   /* 1 (default): MKBF and GUPPI time-first payloads that k_decode_i8_xpose does not take go through k_decode_i8_stage; 0: k_decode_i8_tiled */
#define BB_TUNE_MKBF_CHANNELS 11   /* channels per MKBF tile in k_decode_i8_stage (even, 2..64; default 32) */
#define BB_TUNE_GATHER_CHUNKS 12   /* thread interleave: chunks (floats per thread sample) below this go through the LDS gather kernel; 32 (default) = automatic: every chunk for up to 4 thread slots, chunks below 32 floats otherwise; 4 = only chunks 1 and 2 */
#define BB_TUNE_SEG_TILES     13   /* plain flat kernel: 256-byte tiles per workgroup (default 0 = 32, 16 for 8-bit samples) */
#define BB_TUNE_WORK_STRIPES  18   /* work order of the decode launches: log2 of the number of stripes a launch's work items are dealt over (0 = file order; default -1 = 16 stripes for outputs of 16 GiB and more, 4 below) */
#define BB_TUNE_XPOSE         19   /* 1 (default): int8 transposes with 16-byte aligned input runs go through k_decode_i8_xpose; 0: k_decode_i8_tiled / _stage always */
#define BB_TUNE_XPOSE_ROWS    20   /* k_decode_i8_xpose: tile length, 128 or 64 (output rows of a 64-channel tile); default 0 = 64 for MKBF heaps in 64-channel tiles, 128 otherwise */
#define BB_TUNE_M4_WIDEN      22   /* 1 (default): 16- and 32-track Mark 4 units whose word count and fill prefix allow it are decoded as 64-bit super-words by the 64-track kernels; 0: always the native word size */
#define BB_TUNE_SELECT_BYTES  23   /* payload bytes (all thread slots together) that k_decode_gather_select stages in LDS per work item (256..32768, default 16384) */
#define BB_TUNE_LUT_TILES     24   /* upper bound of 256-byte tiles per wave and work item of the byte-table kernels, stated for 2-bit samples (1..16; half as many for 1-bit, twice as many for 4-bit samples).  Default 0 = by kernel: 4 (32 KiB of output per work item) for k_decode_flat_lut (1- and 4-bit), 6 (48 KiB) for the 2-bit kernel k_decode_flat_lds */
#define BB_TUNE_XPOSE_TC      27   /* k_decode_i8_xpose: channels per tile, 64 / 32 / 16 / 8; default 0 = 32 for channels-first blocks, 16 for time-first blocks, 64 for MKBF heaps, and never wider than the decoded channels need */
#define BB_TUNE_XPOSE_MIN_NC  28   /* blocks decoded whole go through k_decode_i8_xpose from this many channels on (default 8; selections: always from 2) */
#define BB_TUNE_ENCODE_RUNS   30   /* k_encode_flat: runs of 256 float4 a wave takes per step, all loads in flight first: 1 or 2; default 0 = 2 for 4-bit codes, 1 otherwise (the product library builds 2 for 4-bit codes only) */
#define BB_TUNE_M4_TILES      26   /* 64-word tiles per wave and work item of the Mark 4 decode kernels (1..8, default 8) */
#define BB_TUNE_GATHER_GLDS   36   /* the LDS gather kernels stage payload bytes with direct-to-LDS loads (1) or load + ds_write (0); default -1 = by kernel: folded channel subsets (k_decode_gather_select) yes, whole thread interleaves (k_decode_gather) no */
#define BB_TUNE_SELECT_PICK    39   /* folded channel subsets: 1 (default) = k_decode_pick (one work item per wave, direct-to-LDS 16-byte loads) for selections of up to an eighth of a thread sample where its conditions hold, 2 = wherever they hold, 0 = k_decode_gather_select always */
#define BB_TUNE_PICK_BYTES     40   /* k_decode_pick: payload bytes of all thread slots a wave stages per work item (1024..32768, default 4096) */
#define BB_TUNE_VDIF8_LDS_GIB   41   /* VDIF 8-bit frames with contiguous output: GiB of payload from which a launch takes k_decode_flat_lds<8,LDS,glds> (16 tiles per wave) instead of the plain kernel (default 20; 0 = always, 100000 = never) */
#define BB_TUNE_TOUCH_MIB       42   /* the three *_read_window calls read a window of 16 MiB up to this many MiB through once, on the decode's stream, before the decode takes it (default 256 = the memory-side cache; 0 = never) */
#define BB_TUNE_ENCODE_STRIPES 38 /* k_encode_flat: log2 of the number of stripes the 16 KiB input runs of a launch are dealt over (the decode launches' work order, applied to the read stream): 0-10; default 0 = input order */

/* Sets a knob for the CALLING HOST THREAD's later launches (the knobs are
 * thread-local: another thread's launches keep the defaults).  BB_EINVAL for a
 * knob this build does not have. */
int bb_tune(int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* BBDECODE_TUNE_H */
