/*
 * bbdecode.h -- C ABI of libbbdecode.so: MI355X (gfx950) frame-index scan and
 * packed-sample decode for radio-baseband formats (VDIF, Mark 5B, Mark 4,
 * GUPPI, DADA, GSB).
 *
 * This is the drop-in boundary for the hot path of mhvk/baseband
 * (open().read() -> Payload.fromfile -> Payload.data).  Every entry point
 * cites the reference interface (path:line relative to the reference tree)
 * it replaces.  The reference is pure Python; the seam a maintainer would bind
 * is PayloadBase._decoders (base/payload.py:50,314-315) plus the per-frame
 * header reads of the stream readers (base/base.py:919-1125).  See
 * INTEGRATION.md for the ctypes stub.
 *
 * Conventions
 *  - plain pointers and sizes only; all `d_*` pointers are DEVICE pointers
 *    (hipMalloc'ed or equivalent); `stream` is a hipStream_t passed as void*
 *    (NULL = the default stream).
 *  - every function of THIS header is stream-ordered and asynchronous w.r.t.
 *    the host, never allocates and never synchronises (graph-capturable).
 *    State the library keeps: the constant level tables uploaded once per
 *    device by bb_init(); per host thread, the name of the last kernel
 *    (bb_last_kernel), the last HIP error, and the geometry overrides a test or
 *    benchmark sets with bb_tune (bbdecode_tune.h: thread-local, they change
 *    the calling thread's later launches only and never a result).  No
 *    process-wide knobs.  The output arena (bbdecode_arena.h) is a separate
 *    object its creator owns: bb_arena_alloc may map memory and time a probe
 *    launch, i.e. it does synchronise -- see that header.
 *  - return value 0 = success, negative errno-style code otherwise.  No
 *    exceptions cross the ABI.  The Python host maps BB_ENOTSUP to KeyError
 *    (the reference surfaces an unknown coder as KeyError from the _decoders
 *    dict, base/payload.py:314-315).
 *  - decoded output is float32; complex64 is (re, im) interleaved float32.
 */
#ifndef BBDECODE_H
#define BBDECODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BB_ABI_VERSION 7   /* 7 (round 6): bb_touch; the *_read_window calls read small windows through first */   /* 6 (round 6): bb_mark5b_locate_stream */   /* 5 (round 6): bb_arena_stats grew (first_probe_gbps .. second_chance_wins) */   /* 4 (round 5): bb_arena_prepare, bb_arena_owns, bb_arena_stats grew (va_ranges .. prepare_wait_ms); the `reserved` words of the scan parameter blocks got meanings whose zero is the old behaviour (bb_vdif_scan_params.set_nframes, bb_mark5b_/bb_mark4_scan_params.by_position): same layout  */   /* 3 (round 4): bb_copy_frames; bb_arena_stats grew va_reserved / va_used; bb_tune knobs are thread-local */   /* 2: bb_tiled_params grew (npol_stored, pol_first, d_chan_map) */

/* error codes (negative errno values) */
#define BB_OK        0
#define BB_EIO      (-5)   /* a HIP runtime call failed (see bb_last_hip_error) */
#define BB_EINVAL   (-22)  /* bad argument (size, alignment, null pointer) */
#define BB_ERANGE   (-34)  /* output buffer too small / index out of range */
#define BB_ENOTSUP  (-95)  /* unsupported coder / bps / mode (-> KeyError) */

/*
 * Sample coders.  A coder plus bits-per-sample selects the code -> float32
 * level table.
 *   BB_CODER_VDIF    offset binary, LSB-first fields
 *                    (vdif/payload.py:25-103; base/encoding.py:52-56,131-144)
 *                    bps 1: {-1,+1}; 2: {-Hi,-1,+1,+Hi}; 4: (n-8)/2.95;
 *                    8: (n-127.5)/35.5   (true float32 division)
 *   BB_CODER_MARK5B  sign/magnitude (mark5b/payload.py:27-94); also VDIF
 *                    EDV 0xab (vdif/payload.py:151-154)
 *                    bps 1: bit ? -1 : +1; 2: field f -> {-Hi,+1,-1,+Hi}[f]
 *   BB_CODER_INT     two's complement integers cast to float32
 *                    bps 8: int8 (dada/payload.py:13-14, guppi/payload.py:13-14,
 *                    gsb/payload.py:39-42); bps 4: sign-extended nibbles, low
 *                    nibble first (gsb/payload.py:24-36)
 */
enum bb_coder {
    BB_CODER_VDIF   = 0,
    BB_CODER_MARK5B = 1,
    BB_CODER_INT    = 2
};

/* flags in bb_frame_rec.flags */
#define BB_FRAME_OK       0x0001u  /* header matches the stream invariants */
#define BB_FRAME_INVALID  0x0002u  /* frame carries invalid data -> fill_value */

/*
 * One record per frame found by a header scan, in file order.
 * Replaces the (header, file offset) pairs the reference keeps implicitly
 * while walking a file (vdif/frame.py:176-243, base/base.py:1083-1125).
 */
typedef struct bb_frame_rec {
    int64_t  payload_offset;  /* byte offset of the payload in the scanned buffer */
    int32_t  time_index;      /* frame(set) index relative to header0
                                 (vdif/base.py:386-390; mark5b/base.py:206-213) */
    int16_t  thread_id;       /* VDIF thread_id; 0 for other formats */
    uint16_t flags;           /* BB_FRAME_* */
} bb_frame_rec;

/* ---- library ---------------------------------------------------------- */

int         bb_abi_version(void);
const char *bb_strerror(int code);
/* hipError_t of the last failing HIP call on this host thread (0 if none) */
int         bb_last_hip_error(void);
/* Name, template arguments and grid of the decode kernel the calling thread
 * launched last through bb_decode_frames / bb_decode_mark4 /
 * bb_decode_i8_tiled ("" before the first launch).  The dispatcher picks
 * kernels by geometry; benchmarks and profiles report what actually ran
 * instead of assuming it.  The string is thread-local and valid until the
 * thread's next decode call.  (No reference counterpart: the reference's
 * decoders are Python callables, base/payload.py:314-315.) */
const char *bb_last_kernel(void);
/* Upload the constant level tables to the current device.  Called lazily by
 * every launch entry point; exported so hosts can front-load it.  Fails with
 * BB_EIO when no usable gfx950 device/code object is present: there is no
 * CPU fallback in this library. */
int         bb_init(void);
/* Copy the 2^bps-entry code -> level table of (coder, bps) to host memory
 * (what vdif/payload.py:25-63 and mark5b/payload.py:27-72 tabulate per byte). */
int         bb_get_levels(int coder, int bps, float *h_levels, size_t n);
/* The three float32 inputs at which the 2-bit encoder (encode_2bit_base,
 * base/encoding.py:77-102) steps to code 1, 2, 3: code(x) = #{k : x >= thr[k]}.
 * Found on the host by bisection over the reference arithmetic. */
int         bb_get_encode_thresholds(float h_thr[3]);

/* ---- frame index: header scan ------------------------------------------ */

/*
 * VDIF header scan: VDIFHeader.fromfile + verify + _get_index for every frame
 * of a fixed-stride file image (vdif/header.py:158-186,569-589;
 * vdif/base.py:386-390), and the stream-invariant match of
 * base/header.py:588-638 used by locate_frames (base/base.py:181-335).
 * Frame k is expected at first_offset + k*frame_nbytes.
 */
typedef struct bb_vdif_scan_params {
    uint64_t first_offset;    /* byte offset of the first header */
    uint32_t frame_nbytes;    /* header + payload */
    uint32_t header_nbytes;   /* 32, or 16 for legacy headers */
    uint32_t pattern[8];      /* header0 words */
    uint32_t mask[8];         /* 1-bits = stream-invariant header bits */
    int32_t  ref_seconds;     /* header0['seconds'] */
    int32_t  ref_frame_nr;    /* header0['frame_nr'] */
    int32_t  frame_rate;      /* frames per second per thread (integer Hz) */
    int32_t  set_nframes;     /* frames per frame set in file order (threads in the file), 0 or 1: every
                               * frame is placed by its own seconds.  With n > 1 the frames at positions
                               * k*n .. k*n+n-1 from first_offset form a set as VDIFFrameSet.fromfile
                               * builds it (vdif/frame.py:201-243): a frame whose frame_nr equals that of
                               * the set's FIRST frame belongs to the set and takes its time index, whatever
                               * its own seconds say ("we cannot always rely on header['seconds']": VLBA
                               * files whose threads carry different seconds, sample_vlbi.vdif).
                               * bb_vdif_read_window sets it from recs_per_index. */
} bb_vdif_scan_params;

int bb_vdif_scan(const void *d_buf, size_t nbytes,
                 const bb_vdif_scan_params *params,
                 bb_frame_rec *d_recs, size_t nframes, void *stream);

/*
 * Corruption-tolerant frame discovery: byte-granular
 * search for VDIF headers -- the masked-pattern search of
 * VLBIFileReaderBase.locate_frames (base/base.py:181-335) that the reference's
 * _bad_frame recovery relies on (vdif/base.py:536-755).  A position is reported
 * when the stream-invariant pattern matches, the complete frame fits, and
 * another header lies exactly one frame later (for the last frame: earlier).
 * Offsets are appended UNORDERED to d_offsets (capacity `cap`); *d_count
 * (device, caller-zeroed) receives the total number found.  bb_vdif_scan_at()
 * then produces the usual records for explicit, possibly unaligned offsets.
 */
int bb_vdif_locate(const void *d_buf, size_t nbytes,
                   const bb_vdif_scan_params *params,
                   int64_t *d_offsets, size_t cap,
                   unsigned long long *d_count, void *stream);

int bb_vdif_scan_at(const void *d_buf, size_t nbytes,
                    const bb_vdif_scan_params *params,
                    const int64_t *d_offsets, size_t nframes,
                    bb_frame_rec *d_recs, void *stream);

/*
 * Mark 5B header scan (mark5b/header.py:60-68,91-97; mark5b/base.py:206-213):
 * sync word 0xABADDEED, frame_nr, BCD seconds; fixed 10016-byte frames.  The
 * invalid-frame test (payload == 0x11223344 everywhere,
 * mark5b/frame.py:62-70) is folded into the same pass.
 */
typedef struct bb_mark5b_scan_params {
    uint64_t first_offset;
    int32_t  ref_seconds;     /* header0 BCD jday*86400+seconds, decoded */
    int32_t  ref_frame_nr;
    int32_t  frame_rate;
    int32_t  by_position;     /* nonzero: headers are not looked at -- every whole frame counts and is placed
                               * by where it lies (frame k of the call = time index k): a reader with
                               * verify=False, which in the reference reads the frame at the position of an
                               * index and checks nothing (base/base.py:1003-1010) */
} bb_mark5b_scan_params;

int bb_mark5b_scan(const void *d_buf, size_t nbytes,
                   const bb_mark5b_scan_params *params,
                   bb_frame_rec *d_recs, size_t nframes, void *stream);

/*
 * Corruption-tolerant Mark 5B indexing (reference:
 * VLBIStreamReaderBase._bad_frame, base/base.py:1127-1219, with
 * Mark5BFileReader.find_header, mark5b/base.py:136-155, and locate_frames,
 * base/base.py:181-335).  bb_mark5b_locate tests EVERY byte offset: a frame
 * starts at p when the sync word 0xABADDEED sits there, the 10016-byte frame
 * fits in the buffer, the BCD time code passes its CRC-16, and -- when four
 * bytes still fit there -- another sync word sits one frame later (check=1).
 * Offsets are appended unordered to d_offsets (at most `cap`; *d_count, which
 * the caller zeroes, receives the number found).  bb_mark5b_scan_at is
 * bb_mark5b_scan for frames at explicit, possibly odd, offsets
 * (params->first_offset is ignored).
 */
/*
 * Reads [d_buf, d_buf + nbytes) once, with plain loads, and keeps nothing (round 6).  What a
 * launch reads that way stays in the device's 256 MiB memory-side cache for the launches right
 * behind it -- the nontemporal stores of a decode do not push it out -- so a decode whose input
 * was just read through takes it from there: the decode launch of 2^13-2^15 frames of 8 KiB runs
 * 11-16 % faster (profiles/r06cp_exp_tiles_by_size_same_window.log against r06cq_).  The three
 * *_read_window calls do this themselves for windows of 16-256 MiB, on the decode's stream,
 * next to the scan on its own: a whole read() gains 4-7 % (profiles/r06cx_, r06cy_exp_touch_read.log;
 * BB_TUNE_TOUCH_MIB, include/bbdecode_tune.h).  No reference
 * counterpart: the reference reads a file through the page cache (base/base.py:919-969).
 */
int bb_touch(const void *d_buf, size_t nbytes, void *stream);

int bb_mark5b_locate(const void *d_buf, size_t nbytes, int64_t *d_offsets,
                     size_t cap, unsigned long long *d_count, void *stream);
/* ... for ONE stream: word 1 of the header must also agree with `w1_pattern` under
 * `w1_mask` (the bits header0.invariant_pattern() marks in that word: the user
 * word, mark5b/header.py:70-73), at p and -- when it still fits -- one frame
 * later, as the stream reader's searches do, which hand header0 to locate_frames
 * (base/base.py:1083-1219).  w1_mask = 0: bb_mark5b_locate. */
int bb_mark5b_locate_stream(const void *d_buf, size_t nbytes, uint32_t w1_pattern, uint32_t w1_mask,
                            int64_t *d_offsets, size_t cap, unsigned long long *d_count, void *stream);
int bb_mark5b_scan_at(const void *d_buf, size_t nbytes,
                      const bb_mark5b_scan_params *params,
                      const int64_t *d_offsets, size_t nframes,
                      bb_frame_rec *d_recs, void *stream);

/*
 * Turn scan records into the dense, output-ordered source table the decode
 * kernels consume: d_src[time_index*nslot + slot] = payload offset, or -1 for
 * frames that are missing or flagged invalid (they decode to fill_value:
 * base/frame.py:191-199, vdif/frame.py:79-90).  d_thread_slot maps a VDIF
 * thread_id (0..1023) to its output slot or -1 (thread not selected:
 * vdif/base.py:464-490); NULL means "slot 0 for every frame".
 * Replaces VDIFFrameSet.fromfile's gather-by-thread (vdif/frame.py:176-243)
 * and RawOffsets (base/offsets.py).
 */
/*
 * Verification of scan records (verify=True / 'fix': the checks the reference
 * makes frame by frame in VLBIStreamReaderBase._read_frame, base/base.py:
 * 1083-1125 -- header valid, frame index as expected, and a readable header
 * behind the frame).  Adds to *d_nbad the number of records that are not
 * BB_FRAME_OK or, among the first `nstrict`, whose time_index differs from
 * first_index + i / recs_per_index (recs_per_index = threads per frame set).
 */
int bb_verify_records(const bb_frame_rec *d_recs, size_t nrecs,
                      int32_t first_index, uint32_t recs_per_index,
                      size_t nstrict, uint32_t *d_nbad, void *stream);

int bb_build_index(const bb_frame_rec *d_recs, size_t nrecs,
                   const int16_t *d_thread_slot, int nslot,
                   int64_t *d_src, size_t nframes_out, void *stream);

/* ---- packed-sample decode ---------------------------------------------- */

/*
 * Flat LUT decode of whole frames (V1-V3, V8, V10, M5-1, S1, D1 of
 * SURVEY.md section 8a): replaces lut.take(words.view(u1), axis=0)
 * (vdif/payload.py:69-103, mark5b/payload.py:78-94), decode_8bit
 * (base/encoding.py:131-144), int8 astype (dada/payload.py:13-14),
 * PayloadBase._decode/.view/.reshape (base/payload.py:314-330), the frameset
 * thread interleave (vdif/frame.py:402-434) and the invalid -> fill_value
 * branch (base/frame.py:191-199).
 *
 * For output frame f and slot s the payload at d_buf + d_src[f*nslot+s]
 * (payload_nbytes bytes) is expanded to E = payload_nbytes*8/bps float32
 * values v[e]; value e lands at
 *     d_out[((f*R + e/chunk)*nslot + s)*chunk + e%chunk],  R = E/chunk
 * i.e. rows of `chunk` values (chunk = nchan * (2 if complex)) interleaved
 * over nslot threads.  A source of -1 writes fill (re, im alternating when
 * `complex_data`).  With d_src == NULL frames are taken at
 * src0 + (f*nslot + s)*src_stride (fixed-stride file, all valid; the last one
 * must end inside buf_nbytes, else BB_ERANGE).
 *
 * Bounds: every decode entry point follows an index entry only when the whole
 * unit it names (payload_nbytes here; the stream words of a Mark 4 unit; the
 * kept channels' bytes of a block) lies inside [0, buf_nbytes).  Any other
 * entry -- negative, past the end, a stale or corrupt index -- decodes as
 * fill, exactly like -1; nothing outside the buffer is read.  The reference
 * never returns garbage for bytes a file does not hold either (short read ->
 * EOFError, base/payload.py:135-136).
 */
typedef struct bb_decode_params {
    int32_t  coder;            /* enum bb_coder */
    int32_t  bps;              /* 1, 2, 4, 8 */
    int32_t  chunk;            /* float32 values per (sample, slot) */
    int32_t  nslot;            /* threads interleaved in one output row */
    uint64_t payload_nbytes;   /* per frame and slot */
    int64_t  src0;             /* used when d_src == NULL */
    int64_t  src_stride;       /* used when d_src == NULL */
    int32_t  complex_data;     /* selects (fill_re, fill_im) vs fill_re only */
    float    fill_re;
    float    fill_im;
    int32_t  reserved;
} bb_decode_params;

int bb_decode_frames(const void *d_buf, size_t buf_nbytes,
                     const int64_t *d_src, size_t nframes,
                     const bb_decode_params *params,
                     float *d_out, size_t out_elems, void *stream);

/*
 * bb_decode_frames with a CHANNEL SELECTION folded in: of every thread
 * sample's `chunk` floats only positions d_within[0 .. nwithin) (device array,
 * each 0 <= w < chunk, any order, repeats allowed) are written, in that order:
 *   out[((f * R + r) * nslot + s) * nwithin + k] = sample (f, r, s)[d_within[k]].
 * Replaces what the reference does for a reader `subset` that picks channels
 * (decode the whole frame, then index: base/base.py:706-717 with 957-969;
 * vdif/base.py:519-528) without writing -- and re-reading -- the channels
 * nobody asked for.  `d_src` is required (one offset per frame-slot, -1 =
 * fill); `chunk` must be a power of two.  d_out needs 4-byte alignment only.
 * BB_ENOTSUP when the thread slots do not fit the on-chip staging buffer.
 */
int bb_decode_frames_select(const void *d_buf, size_t buf_nbytes,
                            const int64_t *d_src, size_t nframes,
                            const bb_decode_params *params,
                            const int32_t *d_within, int nwithin,
                            float *d_out, size_t out_elems, void *stream);

/*
 * The argument and geometry checks of bb_decode_frames_select without a launch
 * (no device needed): BB_OK when a launch with these parameters and `nwithin`
 * kept positions would be accepted, else the code it would return (BB_EINVAL:
 * more than 4096 positions, sizes that do not fit; BB_ENOTSUP: a thread sample
 * wider than a work item can hold whole rows of -- 16 tiles of 256 bytes -- or
 * more thread slots than the on-chip staging buffer takes).  Readers ask this
 * when they plan a subset (baseband_amd/base/base.py _plan_channel_select) and
 * keep the reference's decode-then-index order otherwise (base/base.py:706-717).
 * Only coder, bps, chunk, nslot and payload_nbytes of `params` are looked at.
 */
int bb_decode_frames_select_check(const bb_decode_params *params, int nwithin);

/*
 * One window of a VDIF stream read in ONE call: bb_vdif_scan over `nframes`
 * frames at scan->first_offset, bb_build_index for `nsets` frame sets,
 * bb_verify_records (when d_nbad is given; `nstrict` records checked for their
 * time index, `recs_per_index` = threads per set in the file), then
 * bb_decode_frames -- or bb_decode_frames_select when nwithin > 0 -- through
 * that index.  The body of the reference's read loop for the frames of a
 * request (base/base.py:957-967 with 1083-1125; vdif/base.py:386-390,464-490;
 * vdif/frame.py:176-243,402-434).  Same results and error codes as the four
 * calls; what it saves is host time (three library entries and their argument
 * marshalling per read() of a binding: 25 us of 65 through ctypes).
 * `verified`: optional hipEvent_t, recorded behind the verification launch and
 * AHEAD of the decode, so that a host that only needs the verdict does not wait
 * for the decode.  `scan_stream` (round 5; NULL = `stream`): the scan, index
 * and verification launches go on THIS stream, `verified` (then required) is
 * recorded there and `stream` -- which takes the decode -- is made to wait for
 * it.  A reader that decodes request after request from bytes that are in HBM
 * already gets the verdict of request k + 1 while the decode of request k is
 * still running on `stream` (back-to-back reads of 2^15 frames: the host waits
 * 0.1 instead of 0.7 ms per read).  The CALLER orders the rest: the input must
 * be complete as far as `scan_stream` can tell, and d_recs / d_src must not be
 * in use by a decode still running on `stream` (a few sets of scratch, taking
 * turns, and a wait for the decode that read a set last:
 * baseband_amd/kernels.py).  d_recs (nframes records) and d_src
 * (nsets * dec->nslot entries) are scratch the caller provides.
 */
int bb_vdif_read_window(const void *d_buf, size_t nbytes,
                        const bb_vdif_scan_params *scan, size_t nframes,
                        const int16_t *d_thread_slot, size_t nsets,
                        const bb_decode_params *dec,
                        const int32_t *d_within, int nwithin,
                        bb_frame_rec *d_recs, int64_t *d_src,
                        float *d_out, size_t out_elems,
                        uint32_t recs_per_index, size_t nstrict, uint32_t *d_nbad,
                        void *verified, void *scan_stream, void *stream);

/*
 * The same for Mark 5B and Mark 4 (single-thread formats: one record per
 * frame, slot 0): `nframes` headers are scanned -- the readers look one header
 * beyond the request when it was staged (mark5b/base.py:136-155,
 * base/base.py:1083-1125) -- and the first `n` frames are decoded; `nstrict` of
 * the records must carry the expected time index.  Mark 4: `nout` > 0 decodes
 * through shorter bit maps (bb_decode_mark4_select), 0 through the full ones.
 */
int bb_mark5b_read_window(const void *d_buf, size_t nbytes,
                          const bb_mark5b_scan_params *scan, size_t nframes, size_t n,
                          const bb_decode_params *dec,
                          const int32_t *d_within, int nwithin,
                          bb_frame_rec *d_recs, int64_t *d_src,
                          float *d_out, size_t out_elems,
                          size_t nstrict, uint32_t *d_nbad, void *verified, void *scan_stream, void *stream);

/*
 * Fetch a device counter (the d_nbad of bb_verify_records) on `side_stream`
 * once `after` (a hipEvent_t, may be NULL) has happened: the stream waits for
 * the event, copies the counter to *h_value (pinned host memory) and the call
 * returns when that copy is done -- without waiting for work queued on other
 * streams behind the event (the decode of the window that was verified).
 */
int bb_fetch_counter(const uint32_t *d_counter, uint32_t *h_value, void *after, void *side_stream);

/* ---- Mark 4 ------------------------------------------------------------ */

/*
 * Mark 4 header scan (M4-2 of SURVEY.md section 8a): for every
 * ntrack*2500-byte frame at first_offset + k*frame_nbytes, test the sync
 * pattern (stream word 63 all zero, words 64-95 all ones:
 * mark4/header.py:345-373, mark4/base.py:110-166), read the four error flags
 * of all tracks (any set -> BB_FRAME_INVALID: mark4/frame.py:78-87) and the
 * BCD time code of track 0 (mark4/header.py:134-141,198-241) to form the
 * frame index.  Times are handled in quarter-milliseconds since the start of
 * `ref_year`; `frame_qms` is the frame duration (5 .. 640).
 * bb_frame_rec.payload_offset is the FRAME start (bb_decode_mark4 skips the
 * 160 header words itself).
 */
typedef struct bb_mark4_scan_params {
    uint64_t first_offset;
    int32_t  ntrack;          /* 16, 32 or 64 */
    int32_t  ref_year;        /* full year of header0 */
    int64_t  ref_qms;         /* header0 time, quarter-ms since start of ref_year */
    int32_t  frame_qms;       /* frame duration in quarter-ms; 0: index = position */
    int32_t  by_position;     /* nonzero: as for bb_mark5b_scan_params (verify=False) */
} bb_mark4_scan_params;

int bb_mark4_scan(const void *d_buf, size_t nbytes,
                  const bb_mark4_scan_params *params,
                  bb_frame_rec *d_recs, size_t nframes, void *stream);

/*
 * Mark 4 longitudinal (along-track) header check -- BASELINE.json configs[3]
 * "longitudinal-parity branch", SURVEY 8a row M4-x.  The 160 header bits of
 * every track end in a CRC-12 (polynomial 0x180f; mark4/header.py:34-44,
 * CRCStack in base/utils.py:200-248).  For frame k at d_offsets[k] (or, with
 * d_offsets NULL, at first_offset + k * ntrack*2500; offsets need not be word
 * aligned) d_bad_tracks[k] receives a mask with bit t set when track t's header
 * does not divide by the polynomial -- what the reference's
 * `crc12._crc(stream)` leaves non-zero (its test asserts `crc12.check(stream)`,
 * mark4/tests/test_mark4.py:57-58).  The reference never applies this check
 * while reading; neither do the decode entry points: it is an extra report that
 * does not alter decoded samples.  Pinned by tests/golden/mark4_crc_cases.json.
 */
int bb_mark4_header_crc(const void *d_buf, size_t nbytes, int ntrack,
                        const int64_t *d_offsets, int64_t first_offset, size_t nframes,
                        uint64_t *d_bad_tracks, void *stream);

/*
 * Corruption-tolerant Mark 4 indexing (reference: _bad_frame,
 * base/base.py:1127-1219, with Mark4FileReader.locate_frames,
 * mark4/base.py:110-166).  bb_mark4_locate tests EVERY byte offset: a frame
 * starts at p when stream word 63 is zero and words 64..95 are all ones, the
 * ntrack*2500-byte frame fits in the buffer, and -- when its pattern still
 * fits -- the frame one later shows the same pattern (check=1).  Output as
 * for bb_mark5b_locate.  bb_mark4_scan_at is bb_mark4_scan for frames at
 * explicit offsets (need not be word aligned; params->first_offset ignored).
 */
int bb_mark4_locate(const void *d_buf, size_t nbytes, int ntrack,
                    int64_t *d_offsets, size_t cap,
                    unsigned long long *d_count, void *stream);
int bb_mark4_scan_at(const void *d_buf, size_t nbytes,
                     const bb_mark4_scan_params *params,
                     const int64_t *d_offsets, size_t nframes,
                     bb_frame_rec *d_recs, void *stream);

/*
 * Mark 4 track-demultiplexing decode (M4-1, M4-2): replaces the five
 * decoders decode_{2chan_2bit_fanout4, 4chan_2bit_fanout4, 8chan_2bit_fanout2,
 * 8chan_2bit_fanout4, 16chan_2bit_fanout2_ft} (mark4/payload.py:122-288) and
 * the header-overwrite fill of Mark4Frame (mark4/frame.py:185-189,248-258).
 * Every stream word of `ntrack` bits yields ntrack/2 float32 values: output j
 * (= fanout sample t * nchan + channel c) takes its sign from bit sign_bit[j]
 * and its magnitude from bit mag_bit[j] of the word; value = {-Hi,-1,+1,+Hi}
 * [2*sign + magnitude].  The maps are data (tests/golden/mark4_bitmaps.json
 * holds the ones of the reference's five decoders).  Unit f is read at
 * d_src[f] (or src0 + f*src_stride); its first `fill_words` words and whole
 * units with source -1 are written as `fill`.
 */
typedef struct bb_mark4_decode_params {
    int32_t  ntrack;          /* 16, 32 or 64 */
    int32_t  reserved;
    uint64_t nwords;          /* stream words per unit: 20000 (frame) or payload size */
    uint64_t fill_words;      /* 160 for frames, 0 for bare payloads */
    int64_t  src0;
    int64_t  src_stride;
    uint8_t  sign_bit[32];
    uint8_t  mag_bit[32];
    float    fill;
    int32_t  reserved2;
} bb_mark4_decode_params;

int bb_decode_mark4(const void *d_buf, size_t buf_nbytes,
                    const int64_t *d_src, size_t nframes,
                    const bb_mark4_decode_params *params,
                    float *d_out, size_t out_elems, void *stream);

/*
 * bb_decode_mark4 with a CHANNEL SELECTION folded in.  Output j of a stream
 * word is sample j / nchan, channel j % nchan of that word (ntrack/2 =
 * fanout * nchan outputs); a reader `subset` that keeps channels c_0 .. c_{m-1}
 * (base/base.py:706-717 applied after mark4/payload.py:333-342 in the
 * reference) is the same decode with the SHORTER maps
 *     sign_bit'[fo * m + k] = sign_bit[fo * nchan + c_k]   (mag_bit alike)
 * of nout = fanout * m entries: every word then yields `nout` floats and a
 * unit nwords * nout, written contiguously -- the channels nobody asked for
 * are neither written nor re-read.  Only the first `nout` (1 .. 32) entries of
 * params->sign_bit / mag_bit are used.  d_out needs 4-byte alignment (16 for
 * the float4 store path, taken when nwords * nout is a multiple of 4).
 */
int bb_decode_mark4_select(const void *d_buf, size_t buf_nbytes,
                           const int64_t *d_src, size_t nframes,
                           const bb_mark4_decode_params *params, int nout,
                           float *d_out, size_t out_elems, void *stream);

/* One window of a Mark 4 stream read in one call (see bb_mark5b_read_window). */
int bb_mark4_read_window(const void *d_buf, size_t nbytes,
                         const bb_mark4_scan_params *scan, size_t nframes, size_t n,
                         const bb_mark4_decode_params *dec, int nout,
                         bb_frame_rec *d_recs, int64_t *d_src,
                         float *d_out, size_t out_elems,
                         size_t nstrict, uint32_t *d_nbad, void *verified, void *scan_stream, void *stream);

/* ---- float32 samples (extension) ---------------------------------------- */

/*
 * `nframes` runs of `nbytes_per_frame` bytes at src0 + f * src_stride of d_buf
 * -> d_out, contiguous: what "decoding" float32 samples amounts to.  EXTENSION
 * -- the reference has no such decoder: DADAPayload._decoders = {8: ...}
 * (dada/payload.py:40-41), NBIT 32 raises KeyError(32) there; BASELINE.json
 * configs[4] names "DADA float32 passthrough".  Nothing to be bit-exact
 * against except the file's own bytes: parity is unpinned by construction and
 * the test is byte identity.  Sizes, offsets and pointers are multiples of 4
 * (BB_EINVAL otherwise); 16-byte loads and stores when they are multiples of
 * 16 (DADA: 4096-byte headers).  BB_ERANGE when a run ends outside
 * buf_nbytes or the output is too small.  Stream-ordered, no host sync.
 */
int bb_copy_frames(const void *d_buf, size_t buf_nbytes, size_t nframes,
                   uint64_t nbytes_per_frame, int64_t src0, int64_t src_stride,
                   void *d_out, size_t out_nbytes, void *stream);

/* ---- byte-aligned formats with an axis permutation --------------------- */

/*
 * int8 (re, im) -> complex64 decode with the axis permutation of
 *   BB_LAYOUT_GUPPI_CF  GUPPI channels-first payloads, (chan, time, pol) ->
 *                       (time, pol, chan)            (guppi/payload.py:90-96)
 *   BB_LAYOUT_MKBF      MeerKAT beamformer DADA heaps, (heap, pol, chan, 256)
 *                       -> (heap*256, pol, chan)     (dada/payload.py:54-89)
 *   BB_LAYOUT_GUPPI_TF  GUPPI time-first ('SIMPLE') payloads, (time, chan, pol)
 *                       -> (time, pol, chan)         (guppi/payload.py:97-102)
 * (G1 and D1 of SURVEY.md section 8a).  From every frame only times
 * [t_lo, t_hi) are decoded -- this is how the GUPPI OVERLAP is dropped
 * (guppi/base.py:203-225) and how partial reads avoid touching whole blocks.
 * Output: frame f, time t, pol p, chan c at
 *   d_out[(((f*(t_hi-t_lo) + t-t_lo)*npol + p)*nchan + c)*2 + {0,1}].
 * Real-valued byte data needs no permutation and goes through
 * bb_decode_frames(BB_CODER_INT, 8).
 */
enum bb_layout {
    BB_LAYOUT_GUPPI_CF = 0,
    BB_LAYOUT_MKBF     = 1,
    BB_LAYOUT_GUPPI_TF = 2
};

typedef struct bb_tiled_params {
    int32_t  layout;          /* enum bb_layout */
    int32_t  npol;
    int32_t  nchan;           /* channels decoded */
    int32_t  nchan_stored;    /* channels the payload holds, when only the `nchan` starting at the
                                 payload offset are decoded (a reader subset that keeps a channel
                                 range: point src0 / d_src at the first kept channel -- c_lo *
                                 ntime * npol * 2 bytes into a GUPPI_CF payload, c_lo * 512 into an
                                 MKBF heap, c_lo * npol * 2 into a GUPPI_TF time); 0 = nchan */
    uint64_t ntime;           /* complete samples stored per frame */
    uint64_t t_lo, t_hi;      /* local sample range to decode, t_hi <= ntime */
    int64_t  src0;            /* payload offsets when d_src == NULL */
    int64_t  src_stride;
    float    fill_re, fill_im;
    /* A reader subset that keeps a LIST of channels and / or one of two
     * polarisations, folded into the decode (the reference decodes whole blocks
     * and indexes afterwards: base/base.py:706-717 after guppi/payload.py:90-102,
     * dada/payload.py:76-79).  The payload offsets point at the payload START
     * (no channel skip), nchan_stored / npol_stored give the stored counts:
     *   out[f, t, p, c] = stored[f, t, pol_first + p, d_chan_map[c]]
     * for p < npol, c < nchan.  d_chan_map: device array of `nchan` stored-channel
     * numbers (each < nchan_stored; any order, repeats allowed), NULL = channels
     * 0 .. nchan-1.  npol_stored 0 = npol.  Only the 16-byte-aligned fast form
     * takes a selection (payload and buffer 16-byte aligned, even nchan, fixed
     * stride); anything else answers BB_ENOTSUP and the caller decodes whole
     * blocks and indexes, like the reference. */
    int32_t  npol_stored;
    int32_t  pol_first;
    const int32_t *d_chan_map;
} bb_tiled_params;

int bb_decode_i8_tiled(const void *d_buf, size_t buf_nbytes,
                       const int64_t *d_src, size_t nframes,
                       const bb_tiled_params *params,
                       float *d_out, size_t out_elems, void *stream);

/* ---- encoders (write side) ----------------------------------------------- */

/*
 * float32 samples -> packed codes, the inverse of bb_decode_frames for one
 * contiguous run.  Thresholds and rounding follow
 * the reference encoders operation by operation: encode_1bit_base /
 * encode_2bit_base (incl. NumPy's floor_divide) / encode_4bit_base /
 * encode_8bit (base/encoding.py:63-158; vdif/payload.py:77-114), Mark 5B
 * sign/magnitude ordering (mark5b/payload.py:84-106), integer formats
 * (gsb/payload.py:44-52, dada/payload.py:17-18).  Complex data are passed as
 * interleaved (re, im) float32.  nelem must be a multiple of 4 and of 8/bps.
 */
int bb_encode_flat(const float *d_in, size_t nelem, int coder, int bps,
                   void *d_out, size_t out_nbytes, void *stream);

/*
 * Mark 4 track multiplexing (mark4/payload.py:138-300): nwords stream words
 * from nwords * ntrack/2 float32 values laid out (sample, channel); the bit
 * maps are those of bb_decode_mark4.
 */
int bb_encode_mark4(const float *d_in, size_t nwords, int ntrack,
                    const uint8_t sign_bit[32], const uint8_t mag_bit[32],
                    void *d_out, size_t out_nbytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BBDECODE_H */
