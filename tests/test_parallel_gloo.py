"""Multi-GPU sharding logic on CPU: world_size-2 gloo processes.

The scanning rank's dense frame index is broadcast; each rank rebases its
slab of the index to the byte range it stages and decodes only that.  The
decoder stand-in here is the NumPy oracle (the GPU kernels are exercised by
the -m gpu tests); what is under test is the partitioning, the collective and
the offset rebasing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bb_oracle_np as orc
from baseband_amd import synth
from baseband_amd.parallel import frame_slab, broadcast_frame_index, local_index

NSETS, NTHREAD, NCHAN, PN = 23, 4, 2, 256


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make():
    order = [2, 0, 3, 1]
    invalid = [(4, 1), (11, 3), (12, 0)]
    image, h0 = synth.random_vdif(77, NSETS, nthread=NTHREAD, nchan=NCHAN, bps=2,
                                  payload_nbytes=PN, frame_rate=10,
                                  thread_order=order, invalid=invalid)
    return image, h0


def _host_index(image, h0):
    """What bb_vdif_scan + bb_build_index produce, computed on the host."""
    fn = h0.frame_nbytes
    src = np.full(NSETS * NTHREAD, -1, np.int64)
    for k in range(len(image) // fn):
        w = image[k * fn:k * fn + 32].view('<u4')
        if w[0] >> 31:
            continue
        tidx = (int(w[0] & 0x3fffffff) - h0['seconds']) * 10 + int(w[1] & 0xffffff) - h0['frame_nr']
        tid = int(w[3] >> 16) & 0x3ff
        src[tidx * NTHREAD + tid] = k * fn + 32
    return src


def _worker(rank, world, port, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        image, h0 = _make()
        src = torch.from_numpy(_host_index(image, h0)) if rank == 0 else None
        src = broadcast_frame_index(src, NSETS * NTHREAD, src_rank=0)
        lo, hi = frame_slab(NSETS, rank, world)
        local, blo, bhi = local_index(src, lo, hi, NTHREAD, PN)
        staged = image[blo:bhi]                     # the only bytes this rank touches
        spf = h0.samples_per_frame
        out = np.zeros(((hi - lo) * spf, NTHREAD, NCHAN), np.float32)
        loc = local.numpy().reshape(hi - lo, NTHREAD)
        for s in range(hi - lo):
            for t in range(NTHREAD):
                o = loc[s, t]
                if o >= 0:
                    out[s * spf:(s + 1) * spf, t] = orc.decode_flat(
                        staged[o:o + PN], 'vdif', 2).reshape(-1, NCHAN)
        np.save(os.path.join(outdir, 'rank%d.npy' % rank), out)
        t = torch.tensor([float(hi - lo)])
        dist.all_reduce(t)
        assert int(t.item()) == NSETS
    finally:
        dist.destroy_process_group()


def test_frame_slab_partition():
    for n in (0, 1, 7, 23, 1069463):
        for world in (1, 2, 3, 8):
            slabs = [frame_slab(n, r, world) for r in range(world)]
            assert slabs[0][0] == 0 and slabs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
            sizes = [b - a for a, b in slabs]
            assert max(sizes) - min(sizes) <= 1


def test_local_index_rebase():
    src = torch.tensor([100, -1, 356, 612, -1, -1, 868, 1124], dtype=torch.int64)
    local, lo, hi = local_index(src, 1, 3, 2, 256)
    assert lo == 352 and hi == 612 + 256
    assert local.tolist() == [4, 260, -1, -1]
    local, lo, hi = local_index(src, 2, 3, 2, 256)
    assert (lo, hi) == (0, 0) and local.tolist() == [-1, -1]


def test_two_rank_sharded_decode(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / ('rank%d.npy' % r)) for r in range(world)]
    got = np.concatenate(parts)
    image, h0 = _make()
    exp, _ = orc.vdif_read(image, frame_rate=10)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
