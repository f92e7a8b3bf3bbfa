#!/usr/bin/env python3
"""Headline benchmark: decoded Msamples/s + achieved HBM GB/s, VDIF 2-bit.

One step = one pass of the hot path (header scan -> index build -> packed
sample decode) over one synthetic file image that is already resident in HBM:
BASELINE.json configs[1], "synthetic 8 GiB single-thread VDIF, 2-bit real,
1 channel, EDV 0" (8032-byte frames, 32000 samples per frame).  With N > 1
GPUs every rank decodes its own time slab of the same size (weak scaling, no
data-path collective: frames are independent).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events
around the dominant kernel (k_decode_flat) on the launching stream;
`cpu_baseline` times the NumPy restatement of the reference's per-frame read
loop (oracle/, "port") on one host core over a bounded sample -- plus, as extra
keys, the same loop on all cores (forked workers, before the GPU is touched)
and the bare LUT take as the NumPy ceiling.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X spec peak (MI355X_MICROARCH.md)
FRAME_NBYTES = 8032
HEADER_NBYTES = 32
PAYLOAD_NBYTES = 8000
SPF = 32000                     # samples per frame (2-bit, real, 1 channel)
FRAME_RATE = 1000               # frames per second -> 32 MHz sample rate


def make_file_image_on_device(nframes, seed, first_frame, device):
    """cfg2 file image born in HBM: uniform random payload bytes + EDV-0
    headers (seconds / frame_nr incrementing).  Same header words as
    baseband_amd.synth / the reference writer would produce."""
    from baseband_amd.vdif.header import VDIFHeader
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    words_per_frame = FRAME_NBYTES // 4
    img = torch.empty(nframes * words_per_frame, dtype=torch.int32, device=device)
    step = 1 << 28
    for lo in range(0, img.numel(), step):          # bounded temporaries
        hi = min(img.numel(), lo + step)
        img[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g,
                                   device=device, dtype=torch.int64).to(torch.int32)
    h0 = VDIFHeader.fromvalues(edv=0, bps=2, nchan=1, complex_data=False,
                               payload_nbytes=PAYLOAD_NBYTES, station='AA',
                               time=np.datetime64('2020-01-01T00:00:00'))
    w = [int(x) for x in h0.words]
    v = img.view(nframes, words_per_frame)
    idx = torch.arange(first_frame, first_frame + nframes, device=device, dtype=torch.int64)
    v[:, 0] = (w[0] + idx // FRAME_RATE).to(torch.int32)
    v[:, 1] = ((w[1] & 0xff000000) + idx % FRAME_RATE).to(torch.int32)

    def s32(x):
        return x - (1 << 32) if x >= (1 << 31) else x
    v[:, 2] = s32(w[2])
    v[:, 3] = s32(w[3])
    v[:, 4:8] = 0
    return img.view(torch.uint8), h0


def _cpu_worker(args):
    """One process of the all-cores CPU leg: its own slab of cfg2 frames through
    the reference-as-written loop for about `seconds`."""
    seed, nframes, seconds = args
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bb_oracle_np as orc
    from baseband_amd import synth
    image, _ = synth.random_vdif(seed, nframes, payload_nbytes=PAYLOAD_NBYTES,
                                 frame_rate=FRAME_RATE)
    orc.vdif_read(image, frame_rate=FRAME_RATE)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.vdif_read(image, frame_rate=FRAME_RATE)
        reps += 1
    return reps * nframes * SPF, time.perf_counter() - t0


def cpu_baseline(target_seconds=12.0):
    """Reference-as-written loop (NumPy port) on a bounded sample: one core (how
    the reference runs), all host cores over disjoint frame slabs (the
    pickle-to-processes advice of the reference's performance tips), and the
    bare LUT `take` without the per-frame loop as the NumPy ceiling.  Called
    before anything touches the GPU, so that forking workers is safe."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bb_oracle_np as orc
    from baseband_amd import synth
    nframes = 4000
    image, _ = synth.random_vdif(12345, nframes, payload_nbytes=PAYLOAD_NBYTES,
                                 frame_rate=FRAME_RATE)
    orc.vdif_read(image, frame_rate=FRAME_RATE)           # warm (LUT, page faults)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.vdif_read(image, frame_rate=FRAME_RATE)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or reps >= 2000:
            break
    msps = reps * nframes * SPF / dt / 1e6
    result = {"value": round(msps, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
              "sample": "{} x {} frames of the same cfg2 layout ({:.1f} MiB each), "
                        "oracle/bb_oracle_np.vdif_read (per-frame NumPy LUT take loop)"
                        .format(reps, nframes, image.size / 2 ** 20),
              "host": "{} logical cores; numpy {}".format(os.cpu_count(), np.__version__)}
    # bare take: every payload byte of the sample through the 256 x 4 table in
    # one call (no headers, no per-frame Python)
    try:
        payload = np.ascontiguousarray(
            image.reshape(nframes, FRAME_NBYTES)[:, HEADER_NBYTES:]).reshape(-1)
        lut = orc.byte_lut('vdif', 2)
        dest = np.empty((payload.size, 4), np.float32)
        step = PAYLOAD_NBYTES                              # cache-sized pieces are fastest
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            for i in range(0, payload.size, step):
                np.take(lut, payload[i:i + step], axis=0, out=dest[i:i + step], mode='clip')
            n += 1
        result["bare_take"] = {"value": round(n * payload.size * 4 / (time.perf_counter() - t0) / 1e6, 1),
                               "unit": "Msamples/s", "cores": 1,
                               "what": "np.take(lut, payload, out=preallocated) in payload-sized "
                                       "pieces: no headers, no index, no allocation"}
    except Exception as exc:                              # report, never fail the bench
        result["bare_take"] = {"error": repr(exc)}
    # all cores: one forked worker per core, 500-frame slabs (64 MiB of output each)
    try:
        import multiprocessing as mp
        ncore = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        nproc = max(1, min(ncore, 64))
        with mp.get_context('fork').Pool(nproc) as pool:
            parts = pool.map(_cpu_worker, [(1000 + i, 500, 5.0) for i in range(nproc)])
        total = sum(p[0] for p in parts)
        slowest = max(p[1] for p in parts)
        result["all_cores"] = {"value": round(total / slowest / 1e6, 1), "unit": "Msamples/s",
                               "cores": nproc,
                               "what": "{} processes x 500-frame slabs for 5 s each, same loop"
                                       .format(nproc)}
    except Exception as exc:
        result["all_cores"] = {"error": repr(exc)}
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--gib', type=float, default=8.0, help="file image size per GPU")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # the CPU leg runs first: it forks workers, which must happen before this
    # process initialises the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and torch.cuda.device_count() > 0:
        cpu = cpu_baseline()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback).")
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)

    from baseband_amd import kernels, _lib
    from baseband_amd.parallel import frame_slab
    kernels.init()

    nframes = int(args.gib * 2 ** 30) // FRAME_NBYTES
    # time-slab sharding: rank r owns frames [r*nframes, (r+1)*nframes)
    first_frame, _ = frame_slab(nframes * world, rank, world)
    image, h0 = make_file_image_on_device(nframes, 12345 + rank, first_frame, device)
    pattern, mask = h0.invariant_pattern()
    out = torch.empty(nframes * SPF, dtype=torch.float32, device=device)
    bytes_in = nframes * FRAME_NBYTES
    bytes_out = nframes * SPF * 4
    alg_bytes = bytes_in + bytes_out

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(k=None):
        recs = kernels.vdif_scan(image, nframes, FRAME_NBYTES, HEADER_NBYTES, pattern,
                                 mask, h0['seconds'], h0['frame_nr'] + first_frame,
                                 FRAME_RATE)
        src = kernels.build_index(recs, nframes, 1, None)
        if k is not None:
            ev[k][0].record()
        kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2,
                              src=src, out=out)
        if k is not None:
            ev[k][1].record()
        return recs, src

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # sanity spot check (outside the timed region; NOT the parity proof, which
    # lives in tests/): three frames re-expanded on the host from the library's
    # own level table, 4 samples per byte, least significant pair first
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    ok = True
    for f in (0, nframes // 3, nframes - 1):
        raw = image[f * FRAME_NBYTES + HEADER_NBYTES:(f + 1) * FRAME_NBYTES].cpu().numpy()
        got = out[f * SPF:(f + 1) * SPF].cpu().numpy()
        want = lev[(raw[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3].reshape(-1)
        ok &= bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg = sum(kern_ms) / len(kern_ms)
    achieved = alg_bytes / (kern_avg * 1e-3) / 1e9
    total_samples = nframes * SPF * world * args.steps
    value = total_samples / elapsed / 1e6

    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE, separate runs of this same command; tools/summarize_prof.py).
    # Only meaningful for the default 8 GiB workload the counters were taken on.
    traffic, traffic_detail = None, None
    tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    if os.path.exists(tpath) and abs(args.gib - 8.0) < 1e-9:
        try:
            with open(tpath) as f:
                traffic_detail = json.load(f)
            traffic = traffic_detail["hbm_bytes_per_launch"]
        except Exception:
            traffic, traffic_detail = None, None

    line = {
        "metric": "decoded Msamples/s, VDIF 2-bit 1-thread (scan + index + decode, input resident in HBM)",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "cfg2: synthetic {:.3f} GiB per GPU single-thread VDIF, "
                               "2-bit real, 1 channel, EDV 0, 8032-byte frames"
                               .format(bytes_in / 2 ** 30),
                   "frames_per_gpu": nframes, "bytes_in_per_gpu": bytes_in,
                   "bytes_out_per_gpu": bytes_out,
                   "output": "full-size float32 tensor kept in HBM (no slab recycling)",
                   "sharding": "time slabs, one per rank, no collective"},
        "roofline": {"bound": "hbm", "kernel": "k_decode_flat_aln<2, 0, true, 2, 16>",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms_avg": round(kern_avg, 4),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "traffic": traffic, "traffic_detail": traffic_detail},
        "sanity_spot_check": ok,
    }
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
