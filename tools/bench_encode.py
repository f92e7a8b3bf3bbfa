#!/usr/bin/env python3
"""Encode-kernel throughput (float32 resident in HBM -> packed codes).
usage: python tools/bench_encode.py [GiB of float32 input, default 8]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib          # noqa: E402
from baseband_amd.mark4._bitmaps import BITMAPS  # noqa: E402
from tools.bench_formats import timeit          # noqa: E402


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    n = int(gib * 2 ** 30) // 4 // 1024 * 1024
    kernels.init()
    x = torch.randn(n, dtype=torch.float32, device='cuda') * 2.2
    cases = [('vdif', 0, b) for b in (1, 2, 4, 8)] + [('mark5b', 1, 2), ('int', 2, 8)]
    for direct in (0, 1):
        kernels.tune(_lib.TUNE_ENCODE_DIRECT, direct)
        for name, coder, bps in cases:
            if direct and bps != 2:
                continue
            ms = timeit(lambda: kernels.encode_flat(x, coder, bps))
            nb = n * 4 + n * bps // 8
            print(json.dumps(dict(case='k_encode_flat %s %d-bit%s' % (name, bps, ' (direct arithmetic)' if direct else ''),
                                  ms=round(ms, 4), algorithmic_GBps=round(nb / ms / 1e6, 1),
                                  frac_of_8TBps=round(nb / ms / 8e9, 4))), flush=True)
        m = BITMAPS[(8, 2, 4)]
        ms = timeit(lambda: kernels.encode_mark4(x, 64, m['sign_bit'], m['mag_bit']))
        nb = n * 4 + n // 4
        print(json.dumps(dict(case='k_encode_mark4 64 tracks%s' % (' (direct arithmetic)' if direct else ''),
                              ms=round(ms, 4), algorithmic_GBps=round(nb / ms / 1e6, 1),
                              frac_of_8TBps=round(nb / ms / 8e9, 4))), flush=True)
    kernels.tune(_lib.TUNE_ENCODE_DIRECT, 0)


if __name__ == '__main__':
    main()
