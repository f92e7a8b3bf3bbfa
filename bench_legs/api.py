"""Legs through the drop-in API on the headline image: parity digests, api_read, invalid_fill."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

def parity_digests():
    """Bit-exactness verdict per configuration:
    the small reference-written golden files of every format are decoded
    through the drop-in API on this GPU and the sha256 of the decoded array is
    compared with the digest of the REFERENCE's output committed in
    tests/golden/manifest.json (written by oracle/gen_golden.py from the real
    reference).  Outside every timed region; the parity proof proper is
    tests/ (-m gpu)."""
    import hashlib
    import baseband_amd as bb
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest.json')) as f:
        cases = json.load(f)['cases']
    plan = [('sample_vdif', bb.vdif.open, {}),
            ('vdif_cfg2_small', bb.vdif.open, None), ('vdif_cfg3_small', bb.vdif.open, None),
            ('m5b_c16_b2', bb.mark5b.open, 'm5b'), ('m4_t64_f4', bb.mark4.open, 'm4'),
            ('guppi_cf_c64_ov0', bb.guppi.open, {}), ('dada_p2_c4_cplx', bb.dada.open, {})]
    res = {}
    for name, opener, kw in plan:
        c = cases[name]
        if kw is None:
            kw = dict(sample_rate=c['frame_rate'] * c['samples_per_frame'])
        elif kw == 'm5b':
            kw = dict(sample_rate=c['frame_rate'] * c['samples_per_frame'], kday=c['kday'],
                      nchan=c['nchan'], bps=c['bps'])
        elif kw == 'm4':
            kw = dict(sample_rate=c['frame_rate'] * c['samples_per_frame'], ntrack=c['ntrack'],
                      decade=2010, verify=False)
        try:
            with opener(os.path.join(ROOT, 'tests', 'golden', c['file']), 'rs', squeeze=False, **kw) as fh:
                got = fh.read().cpu().numpy()
            digest = hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()
            res[name] = {"shape": list(got.shape), "sha256_matches_reference": digest == c['sha256']}
        except Exception as exc:
            res[name] = {"error": repr(exc)[:200]}
    res["all_match"] = all(v.get("sha256_matches_reference") is True for v in res.values())
    return res



def leg_api_read(args, image, out, kern_ms):
    """The headline image through the drop-in API: a stream reader opened on
    the device tensor, ``read(out=out)`` -- one scan / index / decode launch for
    the whole file (base/base.py:919-969 semantics; resident.py)."""
    from baseband_amd import vdif, _lib
    t_open = time.perf_counter()
    fh = vdif.open(image, 'rs', sample_rate=float(SPF * FRAME_RATE))
    open_ms = (time.perf_counter() - t_open) * 1e3
    assert fh.shape == (out.numel(),), (fh.shape, out.numel())
    fh.read(out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.steps):
        fh.seek(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fh.read(out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    kname = _lib.last_kernel()
    verify = fh.verify
    fh.close()
    ms = float(np.mean(ts))
    return {"call": "baseband_amd.vdif.open(<uint8 device tensor>, 'rs', sample_rate=32e6).read(out=<float32 device tensor>)",
            "ms": round(ms, 4), "ms_min": round(min(ts), 4), "open_ms": round(open_ms, 2),
            "value": round(out.numel() / ms / 1e3, 1), "unit": "Msamples/s",
            "verify": verify, "kernel": kname,
            "ms_over_kernel_leg": round(ms / kern_ms, 4),
            "timing": "host wall clock around read() incl. the verification sync, mean of {} calls".format(args.steps)}


def leg_invalid_fill(args, image, out, nframes, h0, first_frame, kern_ms):
    """SURVEY 8(d) "value distributions": the headline file with the
    `invalid_data` bit (word 0, bit 31) set in 1 % of the frames, through the
    same scan + index + decode step: flagged frames come out as the fill value
    (base/frame.py:191-199, vdif/frame.py:79-90) and the fill path costs
    nothing.  The bits are cleared again afterwards."""
    from baseband_amd import kernels, _lib
    pattern, mask = h0.invariant_pattern()
    g = torch.Generator(device=image.device)
    g.manual_seed(99)
    bad = torch.nonzero(torch.rand(nframes, generator=g, device=image.device) < 0.01).reshape(-1)
    w0 = image.view(torch.int32)[::FRAME_NBYTES // 4]
    w0[bad] |= -2 ** 31
    try:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        src = None
        for k in range(-1, args.steps):
            recs = kernels.vdif_scan(image, nframes, FRAME_NBYTES, HEADER_NBYTES, pattern, mask,
                                     h0['seconds'], h0['frame_nr'] + first_frame, FRAME_RATE)
            src = kernels.build_index(recs, nframes, 1, None)
            if k >= 0:
                ev[k][0].record()
            kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=src, out=out)
            if k >= 0:
                ev[k][1].record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        nbad = int(bad.numel())
        filled = int((src < 0).sum().item())
        # flagged frames are fill, their neighbours are data -- checked HERE, on
        # the output of the timed 1 %-invalid launches, before anything else
        # writes into `out` (base/frame.py:191-199)
        f = int(bad[nbad // 2].item())
        is_fill = bool((out[f * SPF:(f + 1) * SPF] == 0).all().item())
        g_ = f + 1 if f + 1 < nframes and not bool((bad == f + 1).any().item()) else max(0, f - 1)
        assert not bool((bad == g_).any().item()), "no unflagged neighbour to check"
        lev = _lib.get_levels(_lib.CODER_VDIF, 2)
        raw = image[g_ * FRAME_NBYTES + HEADER_NBYTES:(g_ + 1) * FRAME_NBYTES].cpu().numpy()
        neighbour_ok = bool(np.array_equal(out[g_ * SPF:(g_ + 1) * SPF].cpu().numpy().view(np.uint32),
                                           expand_2bit(raw, lev).view(np.uint32)))
        # a data frame is not all zeros (so `is_fill` above is not vacuous)
        neighbour_not_fill = not bool((out[g_ * SPF:(g_ + 1) * SPF] == 0).all().item())
        # every frame invalid: the same kernel with the same store pattern and NO
        # reads -- what this device does write-only ("measured achievable")
        none = torch.full_like(src, -1)
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        for a_, b_ in ev2:
            a_.record()
            kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=none, out=out)
            b_.record()
        torch.cuda.synchronize()
        ms_w = float(np.median([a_.elapsed_time(b_) for a_, b_ in ev2][1:]))
        del none
        all_fill = bool((out[g_ * SPF:(g_ + 1) * SPF] == 0).all().item())
    finally:
        w0[bad] &= 2 ** 31 - 1
    alg = nframes * (FRAME_NBYTES + PAYLOAD_NBYTES * 16)
    return {"what": "the headline step with invalid_data set in 1 % of the frames (scan -> index entry -1 -> fill 0.0)",
            "frames_flagged": nbad, "index_entries_invalid": filled, "kernel": _lib.last_kernel(),
            "kernel_ms_avg": round(ms, 4), "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
            "ms_over_headline_kernel": round(ms / kern_ms, 4),
            "flagged_frame_is_fill": is_fill, "neighbour_frame_is_data": neighbour_ok and neighbour_not_fill,
            "checked": "on the output of the timed 1 %-invalid launches, before the all-invalid launch below",
            "all_frames_invalid": {"what": "the same launch with every index entry -1: the kernel's stores, no reads",
                                   "output_is_fill": all_fill,
                                   "kernel_ms": round(ms_w, 4),
                                   "write_GBps": round(nframes * PAYLOAD_NBYTES * 16 / ms_w / 1e6, 1),
                                   "frac_of_peak": round(nframes * PAYLOAD_NBYTES * 16 / ms_w / 1e6 / HBM_PEAK_GBS, 4),
                                   "headline_kernel_ms_over_this": round(kern_ms / ms_w, 4)}}


