"""Shared constants and helpers of bench.py and its legs (bench_legs/*)."""
import json
import socket
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X spec peak (MI355X_MICROARCH.md)
FRAME_NBYTES = 8032
HEADER_NBYTES = 32
PAYLOAD_NBYTES = 8000
SPF = 32000                     # samples per frame (2-bit, real, 1 channel)
FRAME_RATE = 1000               # frames per second -> 32 MHz sample rate
CFG3_THREADS = 8
CFG3_NCHAN = 16
CFG3_ORDER = (1, 3, 5, 7, 0, 2, 4, 6)       # thread id at disk position p (sample.vdif's order)
CFG3_SET_RATE = 1000


def _s32(x):
    return x - (1 << 32) if x >= (1 << 31) else x


def make_file_image_on_device(nsets, seed, first_set, device, nthread=1, nchan=1,
                              complex_data=False, order=(0,), set_rate=FRAME_RATE, into=None):
    """VDIF file image born in HBM: uniform random payload bytes + EDV-0
    headers (seconds / frame_nr from the frame-set index, thread ids in
    `order`).  Same header words as baseband_amd.synth / the reference writer
    would produce.  `into`: optional uint8 tensor of the right size to fill.
    Returns (uint8 tensor, header0)."""
    from baseband_amd.vdif.header import VDIFHeader
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    words_per_frame = FRAME_NBYTES // 4
    nframes = nsets * nthread
    img = (torch.empty(nframes * words_per_frame, dtype=torch.int32, device=device) if into is None
           else into.view(torch.int32))
    assert img.numel() == nframes * words_per_frame
    step = 1 << 28
    for lo in range(0, img.numel(), step):          # bounded temporaries
        hi = min(img.numel(), lo + step)
        img[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g,
                                   device=device, dtype=torch.int64).to(torch.int32)
    h0 = VDIFHeader.fromvalues(edv=0, bps=2, nchan=nchan, complex_data=complex_data,
                               payload_nbytes=PAYLOAD_NBYTES, station='AA',
                               thread_id=order[0],
                               time=np.datetime64('2020-01-01T00:00:00'))
    w = [int(x) for x in h0.words]
    v = img.view(nsets, nthread, words_per_frame)
    idx = torch.arange(first_set, first_set + nsets, device=device, dtype=torch.int64)[:, None]
    v[:, :, 0] = (w[0] + idx // set_rate).to(torch.int32)
    v[:, :, 1] = ((w[1] & 0xff000000) + idx % set_rate).to(torch.int32)
    v[:, :, 2] = _s32(w[2])
    tid = torch.tensor(list(order), device=device, dtype=torch.int64)[None, :]
    v[:, :, 3] = ((w[3] & ~(0x3ff << 16)) | (tid << 16)).to(torch.int32) if nthread > 1 else _s32(w[3])
    v[:, :, 4:8] = 0
    return img.view(torch.uint8), h0


def empty_with_patience(n, dtype, device, tries=12):
    """``torch.empty`` for the 127.5 GiB output.  The image was allocated just
    before, and an arena that had to try several candidate steps has released up
    to 144 GiB a moment ago: memory the driver is still clearing is not
    allocatable yet (seen with tools/experiments/arena_probe3.cpp), so an out-of-memory here
    is retried for a few seconds before it counts."""
    for k in range(tries):
        try:
            return torch.empty(n, dtype=dtype, device=device)
        except torch.cuda.OutOfMemoryError:
            if k == tries - 1:
                raise
            torch.cuda.empty_cache()
            time.sleep(0.5)


def image_buffer(nbytes, device):
    """Device memory for a file image, allocated the way the package keeps
    file bytes in HBM (`fh.stage()`, the staged copy of a large read):
    `baseband_amd.empty_output(dtype=uint8)` -- arena memory from 1 GiB on,
    torch.empty below or with BB_ARENA=0.  Returns (tensor, "arena" | "torch")."""
    import baseband_amd
    from baseband_amd import arena
    t = baseband_amd.empty_output((int(nbytes),), dtype=torch.uint8, device=device)
    ar = arena.default(device)
    return t, ("arena" if ar is not None and ar.owns(t) else "torch")


def _git_commit():
    try:
        return subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'],
                              capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def _run_group(cmd, cwd, env, timeout):
    """subprocess.run with the child in its own process group, which is killed
    as a whole on timeout: a profiler that stops answering must not leave the
    program it started on the GPU behind (this process is about to allocate
    nearly all of HBM).  Returns an object with returncode / stdout / stderr."""
    import signal
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        out, err = p.communicate()
        raise RuntimeError("timed out after {} s: {}".format(timeout, ' '.join(cmd[:4])))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def timed_launches(fn, reps):
    """Median and mean ms of `fn` (one launch) by HIP events on torch's current
    stream, which is the stream the library launches on (kernels._stream)."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.mean(ts))


def expand_2bit(raw, lev):
    """Host re-expansion of 2-bit VDIF payload bytes from the library's own
    level table (4 samples per byte, least significant pair first): the
    in-bench sanity spot check, NOT the parity proof (that lives in tests/)."""
    return lev[(raw[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3].reshape(-1)

