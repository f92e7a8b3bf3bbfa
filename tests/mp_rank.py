"""One rank of tests/test_multiprocess_gpu.py: a FRESH process (started by
tests/mp_launcher.py) that joins a gloo process group, takes cuda:0 -- every
rank of this test shares the box's one GPU, which RCCL refuses and gloo does
not mind -- and decodes its time slab of each case into its own arena:

    vdif_cfg3_small   `sharded_vdif_read`: rank 0 scans the file on the GPU and
                      builds the dense frame index, ONE broadcast replicates it,
                      every rank rebases its slab and decodes it (north_star)
    others            `sharded_read`: seek + read of the rank's slab, no collective

usage: mp_rank.py <rank> <world> <port> <outdir> <case> [<case> ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    cases = sys.argv[5:]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    os.environ.setdefault('BB_ARENA_GIB', '6')
    import datetime
    import torch
    import torch.distributed as dist
    import baseband_amd as bb
    from baseband_amd import parallel, arena
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest.json')) as f:
        manifest = json.load(f)['cases']
    report = {"rank": rank, "world": world, "pid": os.getpid(), "backend": dist.get_backend(), "cases": {}}
    try:
        for name in cases:
            c = manifest[name]
            path = os.path.join(ROOT, 'tests', 'golden', c['file'])
            rate = c['frame_rate'] * c['samples_per_frame'] if 'frame_rate' in c else None
            if name.startswith('vdif'):
                with bb.vdif.open(path, 'rs', squeeze=False, sample_rate=rate) as fh:
                    data, (a, b) = parallel.sharded_vdif_read(fh)        # ranks from the process group
                    how = "sharded_vdif_read (index broadcast from rank 0)"
            elif name.startswith('m5b'):
                with bb.mark5b.open(path, 'rs', squeeze=False, sample_rate=rate, kday=c['kday'], nchan=c['nchan'],
                                    bps=c['bps']) as fh:
                    data, (a, b) = parallel.sharded_read(fh)
                    how = "sharded_read"
            elif name.startswith('dada'):
                with bb.dada.open(path, 'rs', squeeze=False) as fh:
                    data, (a, b) = parallel.sharded_read(fh)
                    how = "sharded_read"
            elif name.startswith('guppi'):
                with bb.guppi.open(path, 'rs', squeeze=False) as fh:
                    data, (a, b) = parallel.sharded_read(fh)
                    how = "sharded_read"
            else:
                raise ValueError(name)
            assert data.is_cuda and data.shape[0] == b - a
            np.save(os.path.join(outdir, '{}_{}.npy'.format(name, rank)), data.cpu().numpy())
            report["cases"][name] = {"first": a, "last": b, "how": how}
        # a large output too: every process has its own arena on the shared device
        t = bb.empty_output((300 << 20,), dtype=torch.float32)
        ar = arena.default()
        report["arena_block"] = bool(ar is not None and ar.owns(t))
        t.fill_(float(rank + 1))
        report["arena_sum_ok"] = bool(float(t[::4096].sum()) == (rank + 1) * t[::4096].numel())
        del t
        dist.barrier()
    finally:
        dist.destroy_process_group()
    with open(os.path.join(outdir, 'report_{}.json'.format(rank)), 'w') as f:
        json.dump(report, f)


if __name__ == '__main__':
    main()
