"""The other kernels of the library on inputs that decode to the headline output size."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

def leg_locate(image, h0, nframes, reps=5):
    """The corruption-tolerant frame search (base/base.py:181-335
    `locate_frames` as the `_bad_frame` recoveries use it) over the whole
    headline image: a read-only sweep that tests EVERY byte position against
    the header pattern and confirms hits one frame later (bb_vdif_locate,
    k_scan.h).  Algorithmic bytes = the file, read once."""
    import ctypes as C
    from baseband_amd import kernels, _lib
    pattern, mask = h0.invariant_pattern()
    p = kernels._vdif_params(FRAME_NBYTES, HEADER_NBYTES, pattern, mask, 0, 0, 0)
    nbytes = image.numel()
    cap = nbytes // FRAME_NBYTES + 16
    offs = torch.empty(cap, dtype=torch.int64, device=image.device)
    count = torch.zeros(1, dtype=torch.int64, device=image.device)

    def run():
        count.zero_()
        _lib.check(_lib.lib.bb_vdif_locate(image.data_ptr(), nbytes, C.byref(p), offs.data_ptr(), cap,
                                           count.data_ptr(), kernels._stream(image)), 'bb_vdif_locate')
    med, mean = timed_launches(run, reps)
    n = int(count.item())
    found = torch.sort(offs[:min(n, cap)]).values
    ok = n == nframes and bool((found == torch.arange(nframes, device=image.device, dtype=torch.int64)
                                * FRAME_NBYTES).all().item())
    return {"case": "bb_vdif_locate: byte-granular header search over the {:.3f} GiB cfg2 image".format(nbytes / 2 ** 30),
            "kernel": "k_vdif_locate (bb_locate_sweep)", "ms": round(mean, 4), "ms_median": round(med, 4),
            "timing": "incl. the memset of the hit counter",
            "algorithmic_GBps": round(nbytes / mean / 1e6, 1), "frac": round(nbytes / mean / 1e6 / HBM_PEAK_GBS, 4),
            "bytes_in": nbytes, "bytes_out": n * 8, "frames_found": n, "spot_check": ok,
            "spot_check_what": "every frame of the image found, at its offset, nothing else"}


def leg_other_configs(device, out, gib=8.0, gib8=31.0, reps=5, res=None):
    """Kernel-level figures for the other BASELINE configurations on random
    input -- `gib` GiB for the 2-bit formats, `gib8` for the 8-bit ones, i.e. the
    same 128-137 GB of decoded output as the headline launch each (the output
    of a launch should span as much of HBM as the headline's does: DESIGN.md 3.1,
    docs/DESIGN_rounds1-3.md "Where the output lies"): (ms, algorithmic GB/s, fraction of 8 TB/s,
    kernel as named by the library)."""
    from baseband_amd import kernels, _lib
    from baseband_amd.mark4._bitmaps import BITMAPS
    nbytes = int(gib * 2 ** 30)
    nbytes8 = int(min(gib8 * 2 ** 30, out.numel()))
    g = torch.Generator(device=device)
    g.manual_seed(4242)
    buf = torch.empty(max(nbytes, nbytes8) + 4096, dtype=torch.uint8, device=device)
    for lo in range(0, buf.numel() // 4, 1 << 28):
        hi = min(buf.numel() // 4, lo + (1 << 28))
        buf.view(torch.int32)[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g,
                                                     device=device, dtype=torch.int64).to(torch.int32)
    res = [] if res is None else res            # (the caller's list keeps the rows measured before a failure)

    def add(name, fn, bytes_in, bytes_out, units, unit_name):
        med, mean = timed_launches(fn, reps)
        gbs = (bytes_in + bytes_out) / mean / 1e6
        res.append({"case": name, "kernel": _lib.last_kernel(), "ms": round(mean, 4),
                    "ms_median": round(med, 4), "algorithmic_GBps": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_in": bytes_in, "bytes_out": bytes_out,
                    "M{}_per_s".format(unit_name): round(units / mean / 1e3, 1)})

    # cfg0 layout: 8 threads x 1 channel 2-bit real, 5032-byte frames (sample.vdif)
    fn_, pn, nth = 5032, 5000, 8
    nsets = min(nbytes // (fn_ * nth), out.numel() // (nth * pn * 4))
    perm = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=device)
    pos = torch.arange(nsets, device=device, dtype=torch.int64)[:, None] * nth + perm[None, :]
    src = (pos * fn_ + 32).reshape(-1).contiguous()
    o = out[:nsets * nth * pn * 4]
    add("VDIF 8 threads x 1 channel 2-bit real (sample.vdif layout)",
        lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=1, nslot=nth, src=src, out=o),
        nsets * nth * fn_, o.numel() * 4, o.numel(), "samples")
    # cfg4a: Mark 5B 16 channels 2-bit
    nfr = min(nbytes // 10016, out.numel() // 40000)
    o = out[:nfr * 40000]
    add("Mark 5B 16 channels 2-bit",
        lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16, src0=16,
                                      src_stride=10016, out=o),
        nfr * 10016, o.numel() * 4, o.numel(), "samples")
    # cfg4b: Mark 4 64 tracks fanout 4
    m = BITMAPS[(8, 2, 4)]
    nfr = min(nbytes // 160000, out.numel() // (20000 * 32))
    o = out[:nfr * 20000 * 32]
    add("Mark 4 64 tracks fanout 4 (8 channels 2-bit)",
        lambda: kernels.decode_mark4(buf, nfr, 64, 20000, m['sign_bit'], m['mag_bit'], fill_words=160,
                                     src0=0, src_stride=160000, out=o),
        nfr * 160000, o.numel() * 4, o.numel(), "samples")
    # cfg5a: GUPPI 8-bit 2 pol complex 64 channels, channels first, 128 MiB blocks
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nbytes = nbytes8
    nfr = max(1, nbytes // blk)
    nb = nfr * T * npol * nchan * 2
    o = out[:nb]
    add("GUPPI 8-bit 2 pol 64 channels, channels first, OVERLAP 0",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0,
                                        src_stride=blk, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    add("GUPPI 8-bit 2 pol 64 channels, time first",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nchan, T, 0, T, src0=0,
                                        src_stride=blk, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    # cfg5b: DADA 8-bit 2 pol complex (flat int8) and MKBF heaps
    nb = nbytes // 4 * 4
    o = out[:nb]
    add("DADA 8-bit 2 pol complex (flat int8 -> float32)",
        lambda: kernels.decode_frames(buf, 1, nb, _lib.CODER_INT, 8, src0=0, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    nheap_t = 64
    Tm = 256 * nheap_t
    blkm = Tm * npol * nchan * 2
    nfr = max(1, nbytes // blkm)
    nb = nfr * blkm
    o = out[:nb]
    add("DADA MKBF heaps 2 pol 64 channels (256-sample heaps)",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_MKBF, npol, nchan, Tm, 0, Tm, src0=0,
                                        src_stride=blkm, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    # cfg5 "DADA float32 passthrough": NBIT 32 is an EXTENSION of this package
    # (the reference raises KeyError(32), dada/payload.py:40-41; parity is
    # unpinned by construction): the reader's `_decode_window` is ONE strided
    # copy launch of the library (bb_copy_frames, csrc/k_copy.h); here 128 MiB
    # payloads behind 4096-byte headers
    blk32 = 128 << 20
    nfr = max(1, min(nbytes8 - 4096, out.numel() * 4) // (blk32 + 4096))
    # the output where the reader puts it: a read() result of 1-64 GiB is a block of the
    # output arena (placement.empty_output); a slice of the headline tensor if that fails
    o, o_mem = None, "a slice of the 127.5 GiB headline tensor"
    try:
        import baseband_amd
        from baseband_amd import arena as _ar
        o = baseband_amd.empty_output((nfr * blk32 // 4,), dtype=torch.float32, device=device)
        a_ = _ar.default(device)
        o_mem = "arena block (placement.empty_output, as dada.open().read() allocates it)" \
            if a_ is not None and a_.owns(o) else "torch.empty"
    except Exception:
        o = None
    if o is None:
        o = out[:nfr * blk32 // 4]
    add("DADA NBIT=32 float32 passthrough (extension, parity unpinned: no reference counterpart)",
        lambda: kernels.copy_frames(buf, nfr, blk32, src0=4096, src_stride=blk32 + 4096, out=o),
        nfr * blk32, nfr * blk32, nfr * blk32 // 4, "samples")
    res[-1]["output_memory"] = o_mem
    k = nfr - 1
    res[-1]["spot_check"] = bool(torch.equal(
        o[k * (blk32 // 4):k * (blk32 // 4) + 4096].view(torch.int32),
        buf[4096 + k * (blk32 + 4096):4096 + k * (blk32 + 4096) + 16384].view(torch.int32)))
    del o

    # ---- the secondary kernels -------------------------
    def expand_bits(raw, lev, bps):
        """host re-expansion of packed codes, least significant field first"""
        sh = np.arange(0, 8, bps, dtype=np.uint8)
        return lev[(raw[:, None] >> sh) & ((1 << bps) - 1)].reshape(-1)

    def flat_case(name, coder, bps, frame, pay, hdr, limit):
        nfr_ = min(int(limit) // frame, out.numel() // (pay * 8 // bps))
        o_ = out[:nfr_ * (pay * 8 // bps)]
        add(name, lambda: kernels.decode_frames(buf, nfr_, pay, coder, bps, src0=hdr, src_stride=frame, out=o_),
            nfr_ * frame, o_.numel() * 4, o_.numel(), "samples")
        f = nfr_ - 1
        raw = buf[f * frame + hdr:f * frame + hdr + pay].cpu().numpy()
        got = o_[f * (pay * 8 // bps):(f + 1) * (pay * 8 // bps)].cpu().numpy()
        res[-1]["spot_check"] = bool(np.array_equal(
            got.view(np.uint32), expand_bits(raw, _lib.get_levels(coder, bps), bps).view(np.uint32)))

    flat_case("VDIF 1-bit real 1 channel, 8032-byte frames", _lib.CODER_VDIF, 1, 8032, 8000, 32, gib * 2 ** 30)
    flat_case("VDIF 4-bit real 1 channel, 8032-byte frames", _lib.CODER_VDIF, 4, 8032, 8000, 32, 2 * gib * 2 ** 30)
    flat_case("VDIF 8-bit real 1 channel, 8032-byte frames", _lib.CODER_VDIF, 8, 8032, 8000, 32, nbytes8)
    flat_case("GSB rawdump 4-bit real (2^22-byte blocks, no headers)", _lib.CODER_INT, 4, 1 << 22, 1 << 22, 0,
              2 * gib * 2 ** 30)
    # a reader `subset` of 2 of 16 channels folded into the decode of 8-thread
    # 16-channel complex VDIF (k_decode_gather_select); bytes moved = every frame
    # read + the kept channels written
    nth, nch, pn, fn_ = 8, 16, 8000, 8032
    nsets = int(gib * 2 ** 30) // (fn_ * nth)
    src = (torch.arange(nsets * nth, device=device, dtype=torch.int64) * fn_ + 32).contiguous()
    within = torch.tensor([6, 7, 24, 25], dtype=torch.int32, device=device)          # channels 3 and 12 (re, im)
    spf = pn * 4 // (2 * nch)
    o = out[:nsets * spf * nth * 4]
    add("VDIF 8 threads x 16 channels 2-bit complex, subset of 2 of 16 channels folded into the decode",
        lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth, src=src,
                                      complex_data=True, out=o, within=within),
        nsets * nth * fn_, o.numel() * 4, o.numel() // 2, "complex_samples")
    full = kernels.decode_frames(buf, 1, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth,
                                 src=src[(nsets - 1) * nth:], complex_data=True)
    res[-1]["spot_check"] = bool(torch.equal(
        full.view(spf, nth, 2 * nch)[:, :, within.long()].reshape(-1).view(torch.int32),
        o[(nsets - 1) * spf * nth * 4:].view(torch.int32)))
    del full
    # a channel LIST (8 scattered of 64) of time-first GUPPI blocks
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nfr = max(1, int(gib * 2 ** 30) // blk)
    cmap = torch.tensor([1, 5, 9, 20, 33, 40, 41, 63], dtype=torch.int32, device=device)
    nsel = int(cmap.numel())
    o = out[:nfr * T * npol * nsel * 2]
    add("GUPPI 8-bit 2 pol 64 channels, time first, channel list of 8 of 64 (bytes moved: every block read, kept channels written)",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nsel, T, 0, T, src0=0, src_stride=blk,
                                        out=o, nchan_stored=nchan, npol_stored=npol, chan_map=cmap),
        nfr * blk, o.numel() * 4, o.numel() // 2, "complex_samples")
    full = kernels.decode_i8_tiled(buf, 1, _lib.LAYOUT_GUPPI_TF, npol, nchan, 4096, 0, 4096, src0=(nfr - 1) * blk,
                                   src_stride=blk)
    res[-1]["spot_check"] = bool(torch.equal(
        full.view(4096, npol, nchan, 2)[:, :, cmap.long()].reshape(-1).view(torch.int32),
        o[(nfr - 1) * T * npol * nsel * 2:][:4096 * npol * nsel * 2].view(torch.int32)))
    del full
    # the encoders: float32 -> packed codes; 4 B read per sample
    from baseband_amd._lib import lib as _L, check as _check
    for bps_, coder_ in ((2, _lib.CODER_VDIF), (4, _lib.CODER_VDIF), (8, _lib.CODER_VDIF)):
        nval = min(out.numel(), (buf.numel() - 4096) * 8 // bps_) // 1024 * 1024
        vals = out[:nval]
        packed = buf[:nval * bps_ // 8]

        def enc():
            _check(_L.bb_encode_flat(vals.data_ptr(), nval, coder_, bps_, packed.data_ptr(), packed.numel(),
                                     kernels._stream(vals)), 'bb_encode_flat')
        if bps_ == 2:
            # `out` holds decoded 2-bit levels nowhere in particular by now: fill its
            # head with a decode, so that encode(decode(x)) == x can be checked
            nchk = 4096
            raw0 = buf[:nchk * 8032].clone()
            kernels.decode_frames(raw0, nchk, 8000, _lib.CODER_VDIF, 2, src0=32, src_stride=8032, out=out[:nchk * 32000])
        add("bb_encode_flat VDIF {}-bit (float32 -> packed codes)".format(bps_), enc, nval * 4, nval * bps_ // 8,
            nval, "samples")
        if bps_ == 2:
            res[-1]["spot_check"] = bool(torch.equal(
                packed[:nchk * 8000].view(nchk, 8000), raw0.view(nchk, 8032)[:, 32:]))
            res[-1]["spot_check_what"] = "encode(decode(x)) == x on 4096 payloads"
            del raw0
    return res


