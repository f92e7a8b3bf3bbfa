"""BASELINE configs[2]: 8-thread VDIF sharded by time slab, index broadcast."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

def leg_cfg3(args, rank, world, device, dist, out):
    """BASELINE configs[2]: 8-thread 2-bit complex 16-channel VDIF sharded by
    time slab.  Rank 0 holds the whole file image, scans every header and
    builds the dense (frame set, thread) -> payload offset index; ONE
    broadcast replicates it (RCCL over xGMI under 'nccl'); every rank rebases
    its slab of the index and decodes its own bytes into its own HBM."""
    from baseband_amd import kernels, _lib
    from baseband_amd.parallel import frame_slab, broadcast_frame_index, local_index
    set_nbytes = FRAME_NBYTES * CFG3_THREADS
    nsets = int(args.cfg3_gib * 2 ** 30) // set_nbytes
    nsets_world = nsets * world
    lo, hi = frame_slab(nsets_world, rank, world)
    kw = dict(nthread=CFG3_THREADS, nchan=CFG3_NCHAN, complex_data=True, order=CFG3_ORDER,
              set_rate=CFG3_SET_RATE)
    if rank == 0:
        # the scanning rank holds the whole file (one allocation, filled slab by
        # slab: no concatenation copy next to the 127.5 GiB output buffer)
        sb = nsets * set_nbytes
        # footprint of rank 0 in this leg: the whole file (world x slab) + scan
        # records (16 B per frame) + the index (8 B per frame) + random-fill
        # temporaries (4 GiB at most), next to `out`, which the caller holds.
        # Checked against what the driver reports free, BEFORE allocating: at
        # N = 8 this is 64 GiB beside the 127.5 GiB output
        need = world * sb + nsets_world * CFG3_THREADS * 24 + (4 << 30)
        free_b, total_b = torch.cuda.mem_get_info(device)
        free_b += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        if need > free_b:
            raise RuntimeError("cfg3 leg: rank 0 needs {:.1f} GiB (whole file of {} ranks + index) but {:.1f} GiB "
                               "are free: lower --cfg3-gib".format(need / 2 ** 30, world, free_b / 2 ** 30))
        whole = torch.empty(world * sb, dtype=torch.uint8, device=device)
        for r in range(world):
            _, h0 = make_file_image_on_device(nsets, 777 + r, frame_slab(nsets_world, r, world)[0], device,
                                              into=whole[r * sb:(r + 1) * sb], **kw)
        slab = whole[:sb]
        h0 = make_file_image_on_device(1, 777, 0, torch.device('cpu'), **kw)[1]     # header of set 0
    else:
        slab, h0 = make_file_image_on_device(nsets, 777 + rank, lo, device, **kw)
    pattern, mask = h0.invariant_pattern()
    thread_slot = kernels.thread_slot_map(list(range(CFG3_THREADS)), device)
    chunk = CFG3_NCHAN * 2
    nelem = nsets * CFG3_THREADS * PAYLOAD_NBYTES * 4
    o = out[:nelem]
    nentries = nsets_world * CFG3_THREADS
    coll = []
    dec = []
    kname = [None]

    def step(k=None):
        src = None
        if rank == 0:
            recs = kernels.vdif_scan(whole, nsets_world * CFG3_THREADS, FRAME_NBYTES, HEADER_NBYTES,
                                     pattern, mask, h0['seconds'], h0['frame_nr'], CFG3_SET_RATE)
            src = kernels.build_index(recs, nsets_world, CFG3_THREADS, thread_slot)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        if dist is not None:
            src = broadcast_frame_index(src, nentries, src_rank=0, device=device)
        e[1].record()
        local, byte_lo, byte_hi = local_index(src, lo, hi, CFG3_THREADS, PAYLOAD_NBYTES)
        assert byte_lo >= lo * set_nbytes and byte_hi <= hi * set_nbytes
        local = local + (byte_lo - lo * set_nbytes)             # offsets into this rank's slab image
        e[2].record()
        kernels.decode_frames(slab, nsets, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, chunk=chunk,
                              nslot=CFG3_THREADS, src=local, complex_data=True, out=o)
        e[3].record()
        kname[0] = _lib.last_kernel()
        if k is not None:
            coll.append((e[0], e[1]))
            dec.append((e[2], e[3]))

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
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
    seen = torch.ones(1, device=device)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(seen)
    # Second variant: every rank scans ITS OWN slab and
    # builds its own index -- no collective, no serial work on rank 0 -- so that
    # a measured curve separates the broadcast's cost from rank 0's whole-file
    # scan.  (The north_star's form is the one above; this one needs every
    # rank to know where its slab's first frame set lies in time, which a
    # fixed-rate file gives and a file with missing frames does not.)
    last_local = [None]

    def step_local():
        recs = kernels.vdif_scan(slab, nsets * CFG3_THREADS, FRAME_NBYTES, HEADER_NBYTES, pattern, mask,
                                 h0['seconds'], h0['frame_nr'] + lo, CFG3_SET_RATE)
        src_l = kernels.build_index(recs, nsets, CFG3_THREADS, thread_slot)
        kernels.decode_frames(slab, nsets, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, chunk=chunk,
                              nslot=CFG3_THREADS, src=src_l, complex_data=True, out=o)
        last_local[0] = src_l

    step_local()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_local()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_l = time.perf_counter() - t0
    # the two indices must be the same table
    src_b = None
    if rank == 0:
        src_b = kernels.build_index(
            kernels.vdif_scan(whole, nsets_world * CFG3_THREADS, FRAME_NBYTES, HEADER_NBYTES, pattern, mask,
                              h0['seconds'], h0['frame_nr'], CFG3_SET_RATE), nsets_world, CFG3_THREADS, thread_slot)
    if dist is not None:
        src_b = broadcast_frame_index(src_b, nentries, src_rank=0, device=device)
    loc_b, blo, _ = local_index(src_b, lo, hi, CFG3_THREADS, PAYLOAD_NBYTES)
    same_index = torch.tensor([1.0 if torch.equal(loc_b + (blo - lo * set_nbytes), last_local[0]) else 0.0], device=device)
    if dist is not None:
        t = torch.tensor([elapsed_l], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_l = float(t.item())
        dist.all_reduce(same_index, op=dist.ReduceOp.MIN)
    del src_b, loc_b
    # sanity: set 5 of this slab, every thread, against a host re-expansion
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    ok = True
    spf = PAYLOAD_NBYTES * 4 // (2 * CFG3_NCHAN)                 # complex samples per frame
    got = o.view(nsets, spf, CFG3_THREADS, chunk)[5].cpu().numpy()
    for p, t in enumerate(CFG3_ORDER):
        fo = (5 * CFG3_THREADS + p) * FRAME_NBYTES
        raw = slab[fo + HEADER_NBYTES:fo + FRAME_NBYTES].cpu().numpy()
        want = expand_2bit(raw, lev).reshape(spf, chunk)
        ok &= bool(np.array_equal(got[:, t].view(np.uint32), want.view(np.uint32)))
    coll_ms = float(np.mean([a.elapsed_time(b) for a, b in coll]))
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in dec]))
    alg = nsets * CFG3_THREADS * (FRAME_NBYTES + PAYLOAD_NBYTES * 16)
    achieved = alg / (dec_ms * 1e-3) / 1e9
    ncomplex = nsets * CFG3_THREADS * PAYLOAD_NBYTES * 2        # complex samples x threads x channels
    return {
        "workload": "cfg3: synthetic {:.3f} GiB per GPU 8-thread VDIF, 2-bit complex, 16 channels, "
                    "EDV 0, thread order on disk {}".format(nsets * set_nbytes / 2 ** 30, list(CFG3_ORDER)),
        "rank0_file_GiB": round(world * nsets * set_nbytes / 2 ** 30, 3),
        "rank0_footprint_GiB": round((world * nsets * set_nbytes + nsets_world * CFG3_THREADS * 24) / 2 ** 30, 3),
        "value": round(ncomplex * world * args.steps / elapsed / 1e6, 1),
        "unit": "M complex samples/s (threads x channels counted)",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "step": "rank 0: bb_vdif_scan + bb_build_index over the whole file; broadcast; "
                "every rank: rebase (parallel.local_index) + bb_decode_frames of its slab",
        "collective": {"op": "broadcast of the dense frame index", "bytes": nentries * 8,
                       "ms": round(coll_ms, 4), "ranks_seen": int(seen.item()),
                       "backend": "nccl (RCCL)" if dist is not None else "none (world size 1)"},
        "roofline": {"bound": "hbm", "kernel": kname[0], "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms_avg": round(dec_ms, 4), "algorithmic_bytes_per_launch": alg},
        "rank_local_scan": {
            "what": "the same decode with every rank scanning its own slab and building its own index: "
                    "no collective, no serial whole-file scan on rank 0",
            "value": round(ncomplex * world * args.steps / elapsed_l / 1e6, 1),
            "unit": "M complex samples/s (threads x channels counted)",
            "ms_per_step": round(elapsed_l / args.steps * 1e3, 4),
            "index_equals_broadcast": bool(same_index.item() == 1.0)},
        "sanity_spot_check": ok}


