"""Multi-GPU sharding path on one GPU: index build + broadcast + slab decode."""
import os

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import golden_path, load_expected, bits_equal

pytestmark = pytest.mark.gpu


def test_build_index_and_slab_decode(manifest, tmp_path):
    """Slabs decoded separately (as ranks would) concatenate to the full read;
    thread order on disk is shuffled and some frames are invalid."""
    from baseband_amd import vdif, synth
    from baseband_amd.parallel import sharded_vdif_read
    image, h0 = synth.random_vdif(5, 101, nthread=8, nchan=16, bps=2, complex_data=True,
                                  payload_nbytes=2000, frame_rate=50,
                                  thread_order=[1, 3, 5, 7, 0, 2, 4, 6],
                                  invalid=[(7, 2), (50, 0), (100, 7)])
    p = tmp_path / 'mt.vdif'
    p.write_bytes(image.tobytes())
    exp, _ = orc.vdif_read(image, frame_rate=50)
    with vdif.open(str(p), 'rs', squeeze=False, sample_rate=50 * h0.samples_per_frame) as fh:
        fh.window_bytes = 1 << 20
        src = fh.build_index()
        assert src.shape[0] == 101 * 8
        s = src.cpu().numpy().reshape(101, 8)
        assert s[7, 2] == -1 and s[50, 0] == -1 and s[100, 7] == -1
        assert np.count_nonzero(s < 0) == 3
        # thread 0 is stored 5th in every frame set
        assert s[0, 0] == 4 * h0.frame_nbytes + 32
        for world in (1, 2, 3, 8):
            parts = []
            for rank in range(world):
                data, (a, b) = sharded_vdif_read(fh, rank=rank, world=world, src=src)
                assert data.shape[0] == b - a
                parts.append(data.cpu().numpy())
            assert bits_equal(np.concatenate(parts), exp), world


def test_nccl_world_size_one(manifest, tmp_path):
    """The collective path with the RCCL backend (world size 1 on this box)."""
    import torch
    import torch.distributed as dist
    from baseband_amd import vdif
    from baseband_amd.parallel import sharded_vdif_read
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        case = manifest['vdif_cfg3_small']
        with vdif.open(golden_path(case['file']), 'rs', squeeze=False,
                       sample_rate=case['frame_rate'] * case['samples_per_frame']) as fh:
            data, (a, b) = sharded_vdif_read(fh)
            assert (a, b) == (0, fh.shape[0])
            assert bits_equal(data.cpu().numpy(), load_expected('vdif_cfg3_small'))
        t = torch.ones(4, device='cuda')
        dist.broadcast(t, src=0)
        dist.all_reduce(t)
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fmt,sample,kw', [
    ('vdif', 'samples/sample.vdif', {}),
    ('mark5b', 'samples/sample.m5b', dict(kday=56000, nchan=8, sample_rate=32e6)),
    ('mark4', 'samples/sample.m4', dict(ntrack=64, decade=2010, sample_rate=32e6)),
    ('guppi', 'samples/sample_puppi.raw', {}),
    ('dada', 'samples/sample.dada', {})])
@pytest.mark.parametrize('world', [1, 2, 3])
def test_generic_time_slab_sharding(fmt, sample, kw, world):
    """parallel.sharded_read: the slabs of all ranks tile the stream and equal
    the single-process read, for every format (no collective involved)."""
    import importlib
    import torch
    from baseband_amd import parallel
    mod = importlib.import_module('baseband_amd.' + fmt)
    path = golden_path(sample)
    with mod.open(path, 'rs', **kw) as fh:
        whole = fh.read()
    parts, edges = [], []
    for rank in range(world):
        with mod.open(path, 'rs', **kw) as fh:
            data, (a, b) = parallel.sharded_read(fh, rank, world)
            assert data.shape[0] == b - a
            parts.append(data)
            edges.append((a, b))
    assert edges[0][0] == 0 and edges[-1][1] == whole.shape[0]
    assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    if fmt == 'guppi':
        # a read that STARTS at a frame takes that frame's own first `overlap`
        # samples, a sequential read takes them from the previous frame's tail
        # (the reference's loop, guppi/base.py:270-278); the sample file's copies
        # differ, so those samples are compared against a direct seek + read
        with mod.open(path, 'rs', **kw) as fh:
            for (a, b), part in zip(edges, parts):
                fh.seek(a)
                assert bool((fh.read(b - a) == part).all())
                assert bool((part[64:] == whole[a + 64:b]).all())
    else:
        assert bool((torch.cat(parts) == whole).all())
