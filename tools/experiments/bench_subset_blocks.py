#!/usr/bin/env python3
"""Channel lists and single polarisations of GUPPI blocks (both storage orders)
and MKBF heaps: the selection folded into k_decode_i8_xpose
(bb_tiled_params.d_chan_map / pol_first) against
decode-everything-then-index, on 8 GiB of input in HBM, outputs in the arena.
bytes moved = what the kernel has to read + what it writes."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib, arena          # noqa: E402
from tools.bench_formats import timeit                 # noqa: E402

kernels.init()
dev = torch.device('cuda')
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
ar = arena.Arena(120 << 30)
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nfr = nbytes // blk
full_out = ar.empty(nbytes)                 # 4 bytes out per byte in
for layout, name in ((_lib.LAYOUT_GUPPI_CF, "GUPPI channels first"), (_lib.LAYOUT_GUPPI_TF, "GUPPI time first"),
                     (_lib.LAYOUT_MKBF, "MKBF heaps")):
    def full():
        return kernels.decode_i8_tiled(buf, nfr, layout, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=full_out)
    ms_full = timeit(full, reps=3)
    for label, pf, npk, chans in (("32 of 64 channels (every second)", 0, 2, list(range(0, 64, 2))),
                                  ("8 of 64 channels (scattered)", 0, 2, [3, 9, 17, 18, 30, 41, 55, 60]),
                                  ("one polarisation, all channels", 1, 1, None),
                                  ("one polarisation, 32 of 64 channels", 0, 1, list(range(1, 64, 2))),
                                  ("64 of 64 channels reversed", 0, 2, list(range(63, -1, -1)))):
        cmap = None if chans is None else torch.tensor(chans, dtype=torch.int32, device=dev)
        nc = nchan if chans is None else len(chans)
        n = nfr * T * npk * nc * 2
        out = ar.empty(n)
        ms_sel = timeit(lambda: kernels.decode_i8_tiled(buf, nfr, layout, npk, nc, T, 0, T, src0=0, src_stride=blk, out=out,
                                                        nchan_stored=nchan, npol_stored=npol, pol_first=pf, chan_map=cmap),
                        reps=3)
        kname = _lib.last_kernel().split(' grid')[0]
        kernels.tune(_lib.TUNE_XPOSE_TC, 64)            # round 3's first form: tiles of 64 channels always
        ms_tc64 = timeit(lambda: kernels.decode_i8_tiled(buf, nfr, layout, npk, nc, T, 0, T, src0=0, src_stride=blk, out=out,
                                                         nchan_stored=nchan, npol_stored=npol, pol_first=pf, chan_map=cmap),
                         reps=3)
        kernels.tune(_lib.TUNE_XPOSE_TC, 0)
        idx = torch.arange(nchan, device=dev) if chans is None else torch.tensor(chans, device=dev)

        def two_pass():
            d = full().view(-1, npol, nchan, 2)
            return d[:, pf:pf + npk][:, :, idx].contiguous()
        ms_two = timeit(two_pass, reps=3)
        # bytes the folded kernel must read: channels-first and MKBF store every (pol,) channel as its own run
        # (dropped ones are not read, a dropped pol of channels-first shares cache lines); time-first rows hold everything
        if layout == _lib.LAYOUT_GUPPI_CF:
            read = nbytes * nc // nchan
        elif layout == _lib.LAYOUT_MKBF:
            read = nbytes * nc // nchan * npk // npol
        else:
            read = nbytes
        print(json.dumps(dict(case="{}, {}".format(name, label), kernel=kname, folded_ms=round(ms_sel, 3), folded_tiles_of_64_ms=round(ms_tc64, 3),
                              decode_then_index_ms=round(ms_two, 3), full_decode_ms=round(ms_full, 3),
                              bytes_read=read, bytes_written=n * 4,
                              folded_GBps_moved=round((read + n * 4) / ms_sel / 1e6, 1))), flush=True)
        del out

# narrow blocks decoded whole: tiles as wide as the block (k_decode_i8_xpose<.., TC>) against the general kernel
for layout, name in ((_lib.LAYOUT_GUPPI_CF, "GUPPI channels first"), (_lib.LAYOUT_GUPPI_TF, "GUPPI time first"),
                     (_lib.LAYOUT_MKBF, "MKBF heaps")):
    for nch in (8, 16, 32):
        Tn = blk // (npol * nch * 2)
        res = {}
        for label, knob in (("xpose", 8), ("general", 65)):
            kernels.tune(_lib.TUNE_XPOSE_MIN_NC, knob)
            ms = timeit(lambda: kernels.decode_i8_tiled(buf, nfr, layout, npol, nch, Tn, 0, Tn, src0=0, src_stride=blk,
                                                        out=full_out), reps=3)
            res[label + "_ms"] = round(ms, 3)
            res[label + "_kernel"] = _lib.last_kernel().split(' grid')[0]
            res[label + "_GBps"] = round(5 * nbytes / ms / 1e6, 1)
        kernels.tune(_lib.TUNE_XPOSE_MIN_NC, 8)
        print(json.dumps(dict(case="{}, whole blocks of {} channels".format(name, nch), **res)), flush=True)
