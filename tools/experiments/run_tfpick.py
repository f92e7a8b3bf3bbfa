"""Time-first channel lists: k_decode_i8_tf_pick against k_decode_i8_xpose (forced with
BB_TUNE_XPOSE_TC), 8 GiB of 64-channel 128 MiB blocks; bytes moved = blocks read + kept channels written."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib
kernels.init()
dev = torch.device('cuda', 0)
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nfr = 64
buf = torch.empty(nfr * blk, dtype=torch.uint8, device=dev)
buf.view(torch.int32).random_()
for name, cm, pol in (("8 scattered of 64", [1, 5, 9, 20, 33, 40, 41, 63], None), ("32 of 64 (every other)", list(range(0, 64, 2)), None),
                      ("reversed 64", list(range(63, -1, -1)), None), ("8 of 64, one polarisation", [1, 5, 9, 20, 33, 40, 41, 63], 1)):
    cmap = torch.tensor(cm, dtype=torch.int32, device=dev)
    nsel = len(cm)
    npd = 1 if pol is not None else 2
    out = torch.empty(nfr * T * npd * nsel * 2, dtype=torch.float32, device=dev)
    res = {}
    ref = None
    for label, tc in (("tf_pick", 0), ("xpose_tc64", 64), ("xpose_tc8", 8)):
        kernels.tune(_lib.TUNE_XPOSE_TC, tc)
        ts = []
        for r in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npd, nsel, T, 0, T, src0=0, src_stride=blk, out=out,
                                    nchan_stored=nchan, npol_stored=npol, pol_first=pol or 0, chan_map=cmap)
            b.record(); b.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        dg = int(out.view(torch.int32)[::1031].to(torch.int64).sum().item())
        ref = dg if ref is None else ref
        res[label] = {"ms": round(ms, 3), "TBps_moved": round((nfr * blk + out.numel() * 4) / ms / 1e9, 3),
                      "kernel": _lib.last_kernel().split(' grid')[0], "same_as_first": dg == ref}
    kernels.tune(_lib.TUNE_XPOSE_TC, 0)
    print(json.dumps({"case": name, "result": res}), flush=True)
    del out
