"""The headline launch into an output that lies in the arena (VERDICT r4 weak 10,
next 8): bench.py's headline output is a plain ``torch.empty`` of 127.5 GiB --
above the 64 GiB the readers take from the arena -- so its rate is also a draw
of where that allocation fell.  This leg, run LAST (the plain output is gone by
then), decodes the same image into ONE arena block of the same size and
reports the kernel's rate there."""
import torch

from .common import *          # noqa: F401,F403


def leg_arena_headline(device, gib=8.0, launches=6):
    from baseband_amd import kernels, _lib, arena, placement
    nframes = int(gib * 2 ** 30) // FRAME_NBYTES
    image, image_memory = image_buffer(nframes * FRAME_NBYTES, device)
    image, h0 = make_file_image_on_device(nframes, 12345, 0, device, into=image)
    ar = placement._arena_for(device, True)
    if ar is None:
        return {"skipped": "no arena (BB_ARENA=0 or no VMM)"}
    out = ar.empty((nframes * SPF,), torch.float32)      # (directly: placement.empty_output stops at 64 GiB)
    if out is None:
        return {"skipped": "the arena could not back a {:.1f} GiB block".format(nframes * SPF * 4 / 2 ** 30)}
    try:
        pattern, mask = h0.invariant_pattern()
        recs = kernels.vdif_scan(image, nframes, FRAME_NBYTES, HEADER_NBYTES, pattern, mask, h0['seconds'],
                                 h0['frame_nr'], FRAME_RATE)
        src = kernels.build_index(recs, nframes, 1, None)
        med, mean = timed_launches(lambda: kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2,
                                                                 src=src, out=out), launches)
        alg = nframes * (FRAME_NBYTES + SPF * 4)
        lev = _lib.get_levels(_lib.CODER_VDIF, 2)
        f = nframes - 1
        raw = image[f * FRAME_NBYTES + HEADER_NBYTES:(f + 1) * FRAME_NBYTES].cpu().numpy()
        ok = bool(np.array_equal(out[f * SPF:].cpu().numpy().view(np.uint32), expand_2bit(raw, lev).view(np.uint32)))
        st = ar.stats()
        return {"what": "the headline launch (same image, index, kernel) into one arena block of the output's size",
                "kernel": _lib.last_kernel(), "kernel_ms_avg": round(mean, 4), "kernel_ms_median": round(med, 4),
                "achieved_GBps": round(alg / mean / 1e6, 1), "frac": round(alg / mean / 1e6 / HBM_PEAK_GBS, 4),
                "output_GiB": round(nframes * SPF * 4 / 2 ** 30, 1), "output_in_arena": bool(ar.owns(out)),
                "arena_steps": st['steps'], "arena_bytes_backed": st['bytes_backed'],
                "arena_grow_ms": round(st['grow_ms'], 1), "input_memory": image_memory, "spot_check": ok}
    finally:
        del out
        try:
            ar.trim()
        except Exception:
            pass
