"""Choosing WHERE in HBM an output tensor lies.

The write rate of a decode launch depends on where its output lies in physical
memory more than on anything inside the kernels (the
same launch runs at 5.3 TB/s in one allocation and at 6.4-6.9 in another; the
L2's write-credit stalls towards the memory controllers tell the two apart).
The library cannot place memory, but a caller who decodes into the same
buffer again and again -- a pipeline that reads chunk after chunk with
``fh.read(out=buffer)`` -- can choose among several allocations once:
`empty_output` allocates a few candidates, times a decode-shaped probe launch
on each while all of them are held (so that they lie in different places) and
keeps the fastest.
"""
import torch

from . import _lib, kernels

__all__ = ['empty_output', 'probe_rate']

_FRAME, _PAYLOAD = 8032, 8000           # the probe decodes 2-bit VDIF-like frames


def probe_rate(out, reps=3):
    """TB/s (input + output bytes) of a 2-bit decode launch that fills `out`
    (float32 or complex64 device tensor; it is overwritten)."""
    flat = torch.view_as_real(out).reshape(-1) if out.is_complex() else out.reshape(-1)
    if flat.dtype != torch.float32 or not flat.is_cuda or flat.data_ptr() % 16:
        raise ValueError("a float32 / complex64 device tensor with 16-byte alignment is needed")
    nframes = flat.numel() // (4 * _PAYLOAD)
    if nframes == 0:
        raise ValueError("the tensor is smaller than one probe frame")
    raw = torch.randint(0, 256, (nframes * _FRAME + 256,), dtype=torch.uint8, device=flat.device)
    src = torch.arange(nframes, device=flat.device, dtype=torch.int64) * _FRAME + (_FRAME - _PAYLOAD)
    target = flat[:nframes * 4 * _PAYLOAD]
    best = None
    for k in range(reps + 1):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        kernels.decode_frames(raw, nframes, _PAYLOAD, _lib.CODER_VDIF, 2, src=src, out=target)
        stop.record()
        stop.synchronize()
        ms = start.elapsed_time(stop)
        if k and (best is None or ms < best):           # (the first launch warms up)
            best = ms
    return nframes * (_FRAME + 16 * _PAYLOAD) / best / 1e9


def empty_output(shape, dtype=torch.float32, candidates=3, device='cuda', report=None):
    """An uninitialised device tensor like ``torch.empty(shape, dtype=dtype)``,
    the fastest to decode into of `candidates` allocations (all of them are held
    while they are probed: `candidates` times the size must fit).  `report`, if
    a list, receives the probed rates in TB/s."""
    kernels.require_gpu()
    held, rates = [], []
    for _ in range(max(1, int(candidates))):
        try:
            t = torch.empty(shape, dtype=dtype, device=device)
        except torch.cuda.OutOfMemoryError:
            break
        held.append(t)
        rates.append(probe_rate(t) if candidates > 1 else 0.)
    if not held:
        raise torch.cuda.OutOfMemoryError("no room for an output of shape {}".format(tuple(shape)))
    if report is not None:
        report.extend(rates)
    keep = held[max(range(len(held)), key=rates.__getitem__)]
    del held
    torch.cuda.empty_cache()        # (the losers go back to the driver, not into torch's cache)
    return keep
