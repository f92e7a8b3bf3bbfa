"""Starting the ranks of an N > 1 run, and the CPU rehearsal of that plumbing."""
import subprocess
import tempfile
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

from .cpu import cpu_baseline
from .line import finish

CPU_JSON_ENV = 'BB_BENCH_CPU_BASELINE_JSON'


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N workers with
    torch.distributed.run (this process has not touched the GPU) and pass
    their output through.  The CPU baseline is timed HERE, before the workers
    exist (the host is otherwise idle, no rank waits for it), and handed to
    rank 0 as a file so that the N > 1 line carries `cpu_baseline` too."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    tmp = None
    if not args.no_cpu_baseline and not args.dry_run and os.path.exists('/dev/kfd'):
        try:
            cpu = cpu_baseline()
            cpu["timed_by"] = "the parent of the {} ranks, before they were started".format(args.gpus)
            fd, tmp = tempfile.mkstemp(prefix='bb_cpu_', suffix='.json', dir='/tmp')
            with os.fdopen(fd, 'w') as f:
                json.dump(cpu, f)
            env[CPU_JSON_ENV] = tmp
        except Exception as exc:                    # the bench goes on without it
            print("bench.py: cpu_baseline failed in the parent: {!r}".format(exc), file=sys.stderr)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py')] + argv
    try:
        return subprocess.run(cmd, env=env).returncode
    finally:
        if tmp:
            try:
                os.remove(tmp)
            except OSError:
                pass


def rank0_footprint_gib(cfg3_gib, world):
    """What rank 0 holds in the cfg3 leg beside the output buffer: the whole file (one
    slab per rank), the scan records (16 B per frame) and the dense index (8 B per frame)
    -- bench_legs/cfg3.py checks the same sum (+ 4 GiB of fill temporaries) against free
    memory before it allocates.  At --gpus 8 and the default 8 GiB per rank: 64.2 GiB."""
    set_nbytes = FRAME_NBYTES * CFG3_THREADS
    nsets = int(cfg3_gib * 2 ** 30) // set_nbytes
    return round((world * nsets * set_nbytes + world * nsets * CFG3_THREADS * 24) / 2 ** 30, 3)


# ------------------------------------------------------------------ dry run
def dry_run(args, rank, world):
    """CPU rehearsal of the multi-rank plumbing (tests/test_bench_cli.py): gloo
    rendezvous, slab partition, the index broadcast, barrier + max-over-ranks
    timing, one JSON line from rank 0.  Nothing is decoded and nothing is
    measured: ``value`` is null and ``dry_run`` is true."""
    import torch.distributed as dist
    from baseband_amd.parallel import frame_slab, broadcast_frame_index
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    nsets = 1000 * world
    lo, hi = frame_slab(nsets, rank, world)
    src = torch.arange(nsets * CFG3_THREADS, dtype=torch.int64) * FRAME_NBYTES + HEADER_NBYTES \
        if rank == 0 else None
    t0 = time.perf_counter()
    if world > 1:
        src = broadcast_frame_index(src, nsets * CFG3_THREADS, src_rank=0)
    coll_ms = (time.perf_counter() - t0) * 1e3
    ok = bool((src[lo * CFG3_THREADS:hi * CFG3_THREADS]
               == torch.arange(lo * CFG3_THREADS, hi * CFG3_THREADS) * FRAME_NBYTES + HEADER_NBYTES).all())
    seen = torch.ones(1)
    mine = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    elapsed = mine.clone()
    per_rank = [mine]
    if world > 1:
        dist.barrier()
        dist.all_reduce(seen)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, mine)             # as the real run gathers `kernel_ms_avg` of every rank
    if rank == 0:
        cpu = None
        handed = os.environ.get(CPU_JSON_ENV)
        if handed and os.path.exists(handed):       # what the parent of an N > 1 run timed
            with open(handed) as f:
                cpu = json.load(f)
        rc = finish({
            "metric": "decoded Msamples/s, VDIF 2-bit 1-thread (scan + index + decode, input resident in HBM)",
            "sanity_spot_check": True,              # nothing is decoded in a dry run
            "cpu_baseline": cpu,
            "roofline": {"traffic": None, "traffic_detail": {
                "hbm_bytes_per_launch": None, "reason": "dry run" if world == 1 else "counter passes run at N = 1 only"}},
            "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "float32", "data": "synthetic",
            "dry_run": True, "ranks_seen": int(seen.item()), "slab_of_rank0": [lo, hi],
            "max_over_ranks_s": float(elapsed.item()),
            "per_rank": {"kernel_ms_avg": [round(float(x.item()) * 1e3, 4) for x in per_rank]},
            "cfg3": {"rank0_footprint_GiB": rank0_footprint_gib(args.cfg3_gib, world),
                     "collective": {"bytes": nsets * CFG3_THREADS * 8, "ms": round(coll_ms, 3),
                                    "ranks_seen": int(seen.item()), "backend": "gloo"},
                     "index_ok": ok}}, args.detail)
    else:
        rc = 0
    if world > 1:
        dist.destroy_process_group()
    return rc

