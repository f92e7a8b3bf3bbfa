/*
 * bbdecode_arena.h -- OUTPUT memory for the decode launches (part of
 * libbbdecode.so; optional: every decode entry point takes any device pointer).
 *
 * Why: on MI355X the write rate of a decode-shaped store stream depends on how
 * its output was allocated.  Into a plain allocation (hipMalloc, torch.empty)
 * of 4-34 GB -- the output of an ordinary read(), base/base.py:919-969 in the
 * reference -- the same launch runs at 5.3-5.7 TB/s in most draws and at
 * 6.4-6.8 in some; into memory that was created as chunks with the HIP virtual
 * memory API and mapped into one virtual range it runs at 6.5-6.8 in EVERY
 * draw, for chunks of 2 to 128 MiB alike (same-process A/B, six fresh outputs
 * per size, five processes: profiles/r03f_exp_arena.log, r03g_exp_arena_*.log,
 * r03h_exp_arena_chunk*.log; docs/DESIGN_rounds1-3.md 3.2) -- provided the block's physical
 * memory spans a wide range of the device: blocks of 8 GiB steps taken first
 * thing in a fresh process decode at 5.3 TB/s for good
 * (profiles/r03k_exp_arena_history.log).  The arena is that: a virtual range
 * of `capacity` bytes, backed on demand in steps of at least 48 GiB (less if
 * the device has less free; BB_ARENA_STEP_GIB) by 32 MiB chunks that are mapped
 * in a fixed pseudo-random order (round 6; rounds 3-5: dealt round robin over
 * the step's 1 GiB "teeth") -- the pieces of any block lie scattered over the
 * whole step, whatever order the driver created them in: a 1 GB decode into one
 * physically contiguous gigabyte runs at 5.2 TB/s, into scattered memory at 6.4
 * (profiles/r06l_exp_tooth_rates.log, r06t_exp_deal_orders.log) -- with a
 * first-fit allocator of 32 MiB granules on top.  A wide step is not
 * always enough: in one process every block of a 48 GiB step decoded at 5.5
 * TB/s (profiles/r03n/bench_plain.json).  Since the rate belongs to the memory,
 * a new step is PROBED with a decode-shaped launch (2^16 frames, three
 * launches, about 7 ms).  With BB_ARENA_TRIES = n > 1 a step that probes below
 * BB_ARENA_MIN_GBPS = 6350 is held aside -- unmapped -- while the next candidate
 * is created somewhere else (none once two candidates probe within 4 % of each
 * other) and the
 * fastest stays.  The DEFAULT (BB_ARENA_TRIES unset or 0; round 6) is one
 * candidate, and up to BB_ARENA_MAX_CANDIDATES = 3 in all while the best so far
 * probes below BB_ARENA_RETRY_BELOW_GBPS = 6300 and the first one's memory was
 * cheap to create (under BB_ARENA_CHEAP_MS_PER_GIB = 3 ms per GiB: the driver is
 * not in the middle of wiping freed pages); the fastest stays (`first_probe_gbps`,
 * `second_chances`, `second_chance_wins`, `probe_history` in the statistics).
 * Scattered steps probe at 6.4-6.8 TB/s (profiles/r06[s-z]_bench_line.json); a
 * further candidate costs 0.2-1.9 s (profiles/r04h_prof_arena_grow.log).  Steps
 * grown in the background (bb_arena_prepare) follow the same rule.  Footprint: a
 * step is at most HALF of the device's free memory (behind a 4 GiB margin); at
 * most two steps exist at any moment (the best so far and the one being
 * probed: a slower candidate goes back to the device at once); another
 * candidate is only created while the device reports room for it plus 4 GiB.
 * Memory is taken from the device when a block needs it and goes back with
 * bb_arena_trim (the Python host trims by itself once an arena has held no
 * block for BB_ARENA_IDLE_S = 30 seconds with no reader open).  What growing
 * costs (profiles/r04h_prof_arena_grow.log): creating 48 GiB of memory that was
 * never used in this boot takes 6-11 ms, of memory that was released before --
 * by this or an earlier process -- 1.5-4.6 s on some boxes (the driver clears
 * it on creation; such a step then decodes about 6 % slower) and 7 ms on others; mapping 12 ms; a probe 6-7 ms.
 *
 * SYNCHRONISATION (unlike include/bbdecode.h's entry points): bb_arena_alloc,
 * when it has to grow, creates and maps memory and runs the probe -- launches
 * on the NULL stream and hipEventSynchronize -- under the arena's mutex: it
 * blocks the calling host thread (and other threads allocating from the same
 * arena) for 30 ms to several seconds, and it is not graph-capturable.  An
 * allocation served from free ranges does neither.  bb_arena_trim and
 * bb_arena_destroy unmap memory: the CALLER must have waited for every launch
 * that touches blocks freed earlier (baseband_amd/arena.py synchronises the
 * events it recorded at each free before it calls them).
 *
 * Not stream-ordered: bb_arena_free makes the block available to the next
 * bb_arena_alloc at once; the caller must have ordered its work on the block
 * before freeing or before using the next block (the Python host frees when the
 * last tensor view dies and, when freed memory is handed out on another stream
 * than the one it was used on, makes that stream wait for an event recorded at
 * the free: baseband_amd/arena.py).  Thread safe.
 */
#ifndef BBDECODE_ARENA_H
#define BBDECODE_ARENA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bb_arena bb_arena;

typedef struct bb_arena_stats {
    uint64_t base;           /* device address of the arena's virtual range */
    uint64_t capacity;       /* size of the virtual range (whole GiB) */
    uint64_t bytes_backed;   /* physical memory mapped now */
    uint64_t bytes_in_use;   /* handed out by bb_arena_alloc and not freed */
    uint64_t largest_free;   /* largest block that fits without growing */
    uint64_t bytes_grown;    /* physical memory taken from the device so far ... */
    uint64_t bytes_trimmed;  /* ... and given back by bb_arena_trim */
    uint32_t chunk_bytes;    /* mapping and allocation granule (32 MiB) */
    uint32_t steps;          /* growth steps mapped now */
    uint32_t blocks;         /* live blocks */
    uint32_t probes;         /* candidate steps probed so far */
    double   last_probe_gbps;/* decode rate the probe measured on the step taken last (0: not probed) */
    double   create_ms;      /* wall time of bb_arena_create */
    double   grow_ms;        /* wall time spent growing (create, map, probe), total */
    uint64_t va_reserved;    /* size of the virtual range NEW steps are placed in: [base, base + va_reserved) */
    uint64_t va_used;        /* addresses handed to growth steps so far, all ranges (never reused) */
    uint32_t va_ranges;      /* virtual ranges reserved now (used-up ones stay reserved: their addresses must not come back) */
    uint32_t va_ranges_made; /* ... and reserved so far */
    uint32_t prepares;       /* growth steps started by bb_arena_prepare */
    uint32_t growing;        /* 1 while one of them is on its way */
    double   prepare_ms;     /* their wall time, total (spent on the library's thread) */
    double   prepare_wait_ms;/* time bb_arena_alloc waited for one, total */
    double   first_probe_gbps;/* probe rate of the FIRST candidate of the last growth (= last_probe_gbps unless a second one won) */
    double   last_create_ms; /* wall time creating that first candidate's memory took (large: the driver was wiping pages) */
    uint32_t second_chances; /* growths that tried a second candidate because the first probed slow and was cheap */
    uint32_t second_chance_wins; /* ... and kept the second */
    uint32_t second_chances_no_room; /* growths that wanted another candidate and had no room for it next to the best so far */
    uint32_t probe_history_n;        /* entries of probe_history in use */
    uint16_t probe_history[16];      /* GB/s of every candidate probed (kept or not), oldest first; the last 16 */
} bb_arena_stats;

/* An arena that backs at most `capacity` bytes (rounded up to whole GiB) of the
 * CURRENT device at one time.  Reserves a virtual range many times that size
 * (32 TiB if the runtime allows; BB_ARENA_VA_GIB): addresses are handed to
 * growth steps -- and to every probed candidate -- by a bump pointer and never
 * reused, because new memory mapped at an address that was unmapped a moment
 * ago can receive a kernel's stores at the OLD pages on this runtime
 * (csrc/bb_arena.inc; tools/experiments/va_reuse_probe.cpp shows it for any
 * address that had a mapping before, however long ago); a range that is used
 * up is followed by another one and stays reserved, so the process's address
 * space bounds what an arena can map over its lifetime (about 2,700 steps of
 * 48 GiB).  No physical memory is taken yet.  BB_EINVAL: capacity
 * == 0; BB_EIO: a HIP call failed (no virtual memory management). */
int bb_arena_create(size_t capacity, bb_arena **arena);

/* A block of at least `bytes` (rounded up to whole granules), a whole number of
 * granules from the base (which is 2 MiB aligned at least).
 * Grows by a step of whole GiB that holds the whole block when no free range
 * is large enough (after waiting for a growth bb_arena_prepare started).
 * *d_ptr = NULL and BB_ERANGE when the capacity, the device's memory or the
 * process's address space is exhausted (a used-up virtual range is followed by
 * a new one as long as one can be reserved). */
int bb_arena_alloc(bb_arena *arena, size_t bytes, void **d_ptr);

/* Start taking memory for a block of `bytes` NOW, on a thread of the library,
 * and return at once: the same step bb_arena_alloc(bytes) would grow by (not
 * probed), unless a free range already holds such a block or a growth is on
 * its way.  bb_arena_alloc waits for a growth in flight before it grows
 * itself, so the sequence prepare -> other work -> alloc pays only what is
 * left of the growth.  What this buys depends on WHY growing is slow: creating
 * memory costs 7 ms per 48 GiB unless the driver is still clearing memory that
 * was released a moment ago -- then the creation waits for that, seconds
 * (profiles/r05b_grow_probe.log, r05c_grow_probe2.log).  BB_ERANGE: capacity
 * or device memory exhausted (nothing started). */
int bb_arena_prepare(bb_arena *arena, size_t bytes);

/* 1 if `d_ptr` lies in one of the arena's virtual ranges, else 0. */
int bb_arena_owns(bb_arena *arena, const void *d_ptr);

/* Return a block (the pointer bb_arena_alloc gave).  BB_EINVAL: not a live block. */
int bb_arena_free(bb_arena *arena, void *d_ptr);

/* Give physical memory back to the device: every growth step that holds no
 * live block.  *released (may be NULL) = bytes. */
int bb_arena_trim(bb_arena *arena, size_t *released);

int bb_arena_get_stats(bb_arena *arena, bb_arena_stats *stats);

/* Unmap and release everything.  Live blocks become invalid. */
int bb_arena_destroy(bb_arena *arena);

#ifdef __cplusplus
}
#endif
#endif /* BBDECODE_ARENA_H */
