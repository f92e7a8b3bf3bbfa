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
 *  - every function is stream-ordered and asynchronous w.r.t. the host, never
 *    allocates and never synchronises (graph-capturable), and keeps no
 *    mutable state except the constant level tables uploaded once per device
 *    by bb_init().
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

#define BB_ABI_VERSION 1

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
    int32_t  reserved;
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
    int32_t  reserved;
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
int bb_mark5b_locate(const void *d_buf, size_t nbytes, int64_t *d_offsets,
                     size_t cap, unsigned long long *d_count, void *stream);
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
 * src0 + (f*nslot + s)*src_stride (fixed-stride file, all valid).
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
    int32_t  reserved;
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

/* ---- tuning knobs (performance experiments; results never change) ------ */
#define BB_TUNE_FLAT_VARIANT   0   /* 0 = workgroup per frame, 1 = byte loads (2-bit), 2 = persistent pipelined 4 waves x 8 tiles, 3 = 2 waves x 16 tiles per frame segment, 4 = contiguous output cut in output space (k_decode_flat_span), 5 = as 3 with 256-byte aligned block loads for contiguous output (k_decode_flat_aln; default), 6-9 = explicit write front (k_decode_flat_front; experiment, slower), 10-12 = one pass with 2/4/8 stripes per wave (k_decode_flat_es; experiment, within +-4 % of 0 and 5), 14 = one float4 per thread and stripe (k_decode_flat_elem; experiment) */
#define BB_TUNE_NT_STORES      1   /* 1 = non-temporal stores */
#define BB_TUNE_BLOCKS         2   /* 0 = default grid; >0 = number of workgroups */
#define BB_TUNE_NT_LOADS       3   /* 1 = non-temporal input loads (experiment) */
#define BB_TUNE_TILE_ELEMS     4   /* elements per tile of bb_decode_i8_tiled (default 8192) */
#define BB_TUNE_GATHER_BYTES   6   /* payload bytes of all thread slots staged in LDS per work item of k_decode_gather; 8192 (default) = automatic: 16384, or 4096 for 1-bit data */
#define BB_TUNE_SEG_TILES 13          /* plain flat kernel: 256-byte tiles per workgroup (default 0 = 32, 16 for 8-bit samples) */
#define BB_TUNE_GATHER_CHUNKS 12      /* thread interleave: chunks (floats per thread sample) below this go through the LDS gather kernel; 32 (default) = automatic: every chunk for up to 4 thread slots, chunks below 32 floats otherwise; 4 = only chunks 1 and 2 */
#define BB_TUNE_MKBF_CHANNELS 11      /* channels per MKBF tile in k_decode_i8_stage (even, 2..64; default 32) */
#define BB_TUNE_TILED_STAGE 10        /* 1 (default): MKBF and GUPPI time-first through k_decode_i8_stage; 0: k_decode_i8_tiled */
#define BB_TUNE_LDS_PAD 9             /* experiment: bytes of unused dynamic LDS per workgroup of the aligned flat kernel (caps workgroups per CU); 0 = none (default) */
#define BB_TUNE_TILES_PER_WAVE_8BIT 8 /* the same bound for 8-bit data in the aligned flat kernel (1..32; above 16 selects the 32-tile instantiation) */
#define BB_TUNE_TILES_PER_WAVE 7   /* upper bound of 256-byte tiles per wave and work item in the flat kernels (1..16, default 12) */
#define BB_TUNE_ENCODE_DIRECT  5   /* 1 = 2-bit encoders evaluate the reference clip/add/floor_divide arithmetic per sample instead of comparing with the three thresholds derived from it (check mode) */
#define BB_TUNE_M4_TILES 26           /* 64-word tiles per wave and work item of the Mark 4 decode kernels (1..8, default 8: 4 waves x 8 tiles = 256 KiB of output per work item for 64 tracks) */
#define BB_TUNE_LUT_SMALL 25          /* 1: k_decode_flat_lut instantiated for at most 4 tiles per wave when the work items are that short (experiment: slower); 0 (default): the 16-tile instantiation */
#define BB_TUNE_LUT_TILES 24          /* upper bound of 256-byte tiles per wave and work item in k_decode_flat_lut for 2-bit samples (1..16, default 4; half as many for 1-bit, twice as many for 4-bit samples: 32 KiB of output per work item; its grid is one work item per workgroup up to 2^23 unless BB_TUNE_BLOCKS says otherwise) */
#define BB_TUNE_SELECT_BYTES 23       /* payload bytes (all thread slots together) that k_decode_gather_select stages in LDS per work item (256..32768, default 16384) */
#define BB_TUNE_M4_WIDEN 22           /* 1 (default): 16- and 32-track Mark 4 units whose word count and fill prefix allow it are decoded as 64-bit super-words (4 / 2 stream words per lane and load) by the 64-track kernels; 0: always the native word size */
#define BB_TUNE_BYTE_LUT 21           /* 1 (default): contiguous 1-, 2- and 4-bit decode through the byte table kernel k_decode_flat_lut; 0: k_decode_flat_aln (register level select; the plain kernel for 4-bit) */
#define BB_TUNE_XPOSE_ROWS 20         /* k_decode_i8_xpose: output rows per tile, 128 or 64; default 0 = 128 for GUPPI channels-first blocks and for outputs of 96 GiB and more, 64 for smaller launches of time-first blocks and MKBF heaps */
#define BB_TUNE_XPOSE 19              /* 1 (default): int8 transposes with 16-byte aligned input runs go through k_decode_i8_xpose; 0: k_decode_i8_tiled / _stage always */
#define BB_TUNE_WORK_STRIPES 18       /* work order of the decode launches: log2 of the number of stripes a launch's work items are dealt over (0 = file order; default -1 = 16 stripes for outputs of 16 GiB and more, 4 below) */
#define BB_TUNE_OUT_STRIPE_W 16       /* experiment: deal the frames of a contiguous-output launch over this many output regions (0 = off) ... */
#define BB_TUNE_OUT_STRIPE_S 17       /* ... that lie this many frame-slots apart: frame fs goes to slot (fs % W) * S + fs / W */
#define BB_TUNE_FRONT_GROUP 14        /* k_decode_flat_front (variants 6-9): workgroups per group = width of the write front (default 2048) */
#define BB_TUNE_FRONT_STEPS 15        /* k_decode_flat_front: steps a group sweeps its region in (default 16) */
int bb_tune(int knob, int value);

/* Measurement aid: when d_times is not NULL, the contiguous-output flat decode
 * kernels store the device wall clock (100 MHz ticks) at which each work item
 * was completed into d_times[item] (the caller sizes it for the launch: frames
 * x work items per frame, plus one slot per workgroup of the launch behind
 * them, where the aligned kernel stores its start time).  NULL switches it off (default).  Not for
 * production use: one extra 8-byte store per 32-64 KiB of output. */
int bb_debug_trace(uint64_t *d_times);

/* ---- host staging helpers (measurement only) -------------------------------
 * bb_host_register pins a range of host memory where it lies -- e.g. a window
 * of a read-only file mapping whose pages are in the page cache -- so that
 * bb_copy_to_device (hipMemcpyAsync on `stream`) moves it without a host-side
 * copy into a pinned buffer first; bb_host_unregister releases it after the
 * copy has completed.  Used by tools/exp_hostregister.py: 56-57 GB/s from a
 * populated mapping, but pinning and unpinning the pages of a freshly mapped
 * file costs more host time than the staging copy it would replace
 * (profiles/r02ax_exp_hostregister.log), so the readers keep the pinned
 * double-buffer pipeline (baseband_amd/staging.py). */
int bb_host_register(const void *h_ptr, size_t nbytes);
int bb_host_unregister(const void *h_ptr);
int bb_copy_to_device(void *d_dst, const void *h_src, size_t nbytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BBDECODE_H */
