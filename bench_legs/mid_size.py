"""The launch sizes ordinary read() calls issue, into fresh outputs."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

def leg_mid_size(device, image, draws=5, launches=6):
    """the launch sizes an ordinary ``read()`` issues
    (/root/reference semantics: base/base.py:919-969) -- cfg2 windows of 2^15,
    2^16 and 2^18 frames (4.2, 8.4, 33.6 GB of output) and a GUPPI
    channels-first read of 8 GiB (34 GB of output).  Every draw is a FRESH
    output: `draws` new ``torch.empty`` allocations (the cache emptied in
    between, so each is a new piece of HBM), and `draws` new blocks from the
    placement arena -- what the readers allocate from by default
    (baseband_amd/placement.py) -- taking turns.  Every launch decodes the NEXT
    window of the 8 GiB image (nothing of the input can still be in the 256
    MiB Infinity Cache); a draw's figure is the median of `launches` launches by
    HIP events on the launching stream; reported: min / median / max over the
    draws of the fraction of 8 TB/s.  Also one timed pass through the drop-in
    API per size: ``fh.read(count)`` allocating its own output."""
    import baseband_amd
    from baseband_amd import kernels, _lib, arena, placement, vdif
    img_frames = image.numel() // FRAME_NBYTES
    nxt = [0]
    peak_use = [0]

    def rate(out, nf):
        if ar is not None:
            peak_use[0] = max(peak_use[0], int(ar.stats()["bytes_in_use"]))
        ts = []
        for r in range(launches + 1):
            if nxt[0] + nf > img_frames:
                nxt[0] = 0
            first = nxt[0]
            nxt[0] += nf
            win = image[first * FRAME_NBYTES:(first + nf) * FRAME_NBYTES]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            kernels.decode_frames(win, nf, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=HEADER_NBYTES,
                                  src_stride=FRAME_NBYTES, out=out)
            b.record()
            b.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        return nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / float(np.median(ts)) / 1e6      # GB/s

    def summary(v):
        f = np.array(v) / HBM_PEAK_GBS
        return {"GBps_per_draw": [round(x, 1) for x in v],
                "frac_min": round(float(f.min()), 4), "frac_median": round(float(np.median(f)), 4),
                "frac_max": round(float(f.max()), 4)}

    torch.cuda.empty_cache()
    # the arena the readers create on their first large output (placement.py);
    # registered as an open reader for the length of this leg: an arena whose last
    # block dies while no reader is open gives its memory back, and every draw
    # below would grow (and probe) a new step
    placement.reader_opened()
    ar = placement._arena_for(device)
    res = {"arena": None if ar is None else ar.stats(),
           "method": "fresh output per draw ({} draws, torch.empty and arena blocks taking turns); per draw the median of "
                     "{} launches, each on the next window of the resident 8 GiB image; HIP events".format(draws, launches),
           "sizes": []}
    for lf in (15, 16, 18):
        nf = 1 << lf
        n = nf * SPF
        v_t, v_a, held = [], [], []
        for d in range(draws):
            o = torch.empty(n, dtype=torch.float32, device=device)
            v_t.append(rate(o, nf))
            del o
            torch.cuda.empty_cache()
            o = ar.empty(n) if ar is not None else None
            if o is None and held:
                held.clear()                         # the arena is full of the pieces held back: start over
                o = ar.empty(n)
            if o is not None:
                v_a.append(rate(o, nf))
                held.append(ar.empty((64 << 20) // 4))      # so that the next block starts elsewhere
                del o
        del held
        row = {"frames": nf, "output_GB": round(n * 4 / 1e9, 2), "kernel": _lib.last_kernel(),
               "torch_empty": summary(v_t), "arena": summary(v_a) if v_a else None}
        # the drop-in API on the same image: read(count) allocates its own output
        try:
            with vdif.open(image, 'rs', sample_rate=float(SPF * FRAME_RATE)) as fh:
                ts = []
                for k in range(4):
                    fh.seek(((k * 3 + 1) * nf % (img_frames - nf)) * SPF)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    got = fh.read(nf * SPF)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    inside = ar is not None and ar.owns(got)
                    if k < 3:
                        del got
                ms = float(np.median(ts[1:])) * 1e3
                # the decode kernel alone, into the VERY output the last read() drew (same
                # placement) from the same window: what is left of the call's time is what
                # stands in front of and behind the kernel -- host work, the scan and index
                # launches, the allocation, the verdict's copy
                flat = got.reshape(-1)
                kts = []
                for r in range(5):
                    # (another window every time: 263 MB of input that was decoded a moment ago
                    # would come from the 256 MiB Infinity Cache and flatter the kernel by 6 %)
                    first = ((10 + 3 * r) * nf % (img_frames - nf))
                    win = image[first * FRAME_NBYTES:(first + nf) * FRAME_NBYTES]
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    kernels.decode_frames(win, nf, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=HEADER_NBYTES,
                                          src_stride=FRAME_NBYTES, out=flat)
                    b.record()
                    b.synchronize()
                    if r:
                        kts.append(a.elapsed_time(b))
                kernel_ms = float(np.median(kts))
                del got, flat
                # the same calls back to back, no host sync in between: read()
                # returns once its frames are verified, the decode goes on behind it
                nb2b = 8
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(nb2b):
                    fh.seek(((k * 5 + 2) * nf % (img_frames - nf)) * SPF)
                    got = fh.read(nf * SPF)
                    del got
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                ms_b2b = (time.perf_counter() - t0) / nb2b * 1e3
                row["api_read"] = {"call": "fh.read({} * 32000) at changing offsets, output allocated by the reader".format(nf),
                                   "ms_median": round(ms, 3), "output_in_arena": bool(inside),
                                   "GBps": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / ms / 1e6, 1),
                                   "frac": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / ms / 1e6 / HBM_PEAK_GBS, 4),
                                   "timing": "host wall clock incl. scan, index, allocation and the verification sync",
                                   "kernel_ms_same_output": round(kernel_ms, 3),
                                   "kernel_frac_same_output": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / kernel_ms / 1e6 / HBM_PEAK_GBS, 4),
                                   "host_and_scan_ms": round(ms - kernel_ms, 3),
                                   "back_to_back": {"reads": nb2b, "ms_per_read": round(ms_b2b, 3),
                                                    "host_ms_per_read": round(t_host / nb2b * 1e3, 3),
                                                    "frac": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / ms_b2b / 1e6 / HBM_PEAK_GBS, 4),
                                                    "what": "the same read() calls without a host sync in between"}}
        except Exception as exc:
            row["api_read"] = {"error": repr(exc)[:300]}
        res["sizes"].append(row)
    # GUPPI channels first, 8 GiB in -> 34 GB out (the first 8 GiB of the image as 64 blocks of 128 MiB)
    try:
        npol, nchan, blk = 2, 64, 128 << 20
        T = blk // (npol * nchan * 2)
        nfr = min(64, image.numel() // blk)
        nb = nfr * blk
        v_t, v_a = [], []

        def grate(o):
            med, mean = timed_launches(lambda: kernels.decode_i8_tiled(image, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T,
                                                                       src0=0, src_stride=blk, out=o), launches)
            return (nb + nb * 4) / med / 1e6
        for d in range(draws):
            o = torch.empty(nb, dtype=torch.float32, device=device)
            v_t.append(grate(o))
            del o
            torch.cuda.empty_cache()
            o = ar.empty(nb) if ar is not None else None
            if o is not None:
                v_a.append(grate(o))
                del o
        res["guppi_cf_8GiB_in"] = {"output_GB": round(nb * 4 / 1e9, 2), "kernel": _lib.last_kernel(),
                                   "torch_empty": summary(v_t), "arena": summary(v_a) if v_a else None}
    except Exception as exc:
        res["guppi_cf_8GiB_in"] = {"error": repr(exc)[:300]}
    if ar is not None:
        st = ar.stats()
        res["arena_after"] = st
        res["arena_bytes_backed_per_byte_in_use_peak"] = (
            round(st["bytes_backed"] / max(1, peak_use[0]), 2) if peak_use[0] else None)
        res["arena_peak_bytes_in_use"] = peak_use[0]
    placement.reader_closed()           # (the arena trims itself now: nothing of it is alive)
    if ar is not None:
        res["arena_bytes_backed_after_last_reader_closed"] = ar.stats()["bytes_backed"]
    return res


# ------------------------------------------------------------------------ main
