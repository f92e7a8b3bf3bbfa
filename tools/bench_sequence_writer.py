"""A stream writer over a SEQUENCE of files (DADA: one 128 MiB frame per file;
VDIF: 256 MiB files): GB/s of file bytes with the background sink writing
positionally on 1 / 2 / 4 / 8 threads (BB_WRITE_THREADS) and with the sink off
(profiles/r05g_exp_file_write2.log: several files at once scale).
    python tools/bench_sequence_writer.py"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(fmt):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import baseband_amd as bb
    from baseband_amd.vdif.header import VDIFHeader
    from baseband_amd.dada.header import DADAHeader
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    tmp = tempfile.mkdtemp(prefix='bb_seqw_', dir=os.environ.get('TMPDIR', '/tmp'))
    try:
        if fmt == 'dada':
            h0 = DADAHeader.fromvalues(time=np.datetime64('2013-07-02T01:39:20'), offset=0., sample_rate=16e6, bps=8,
                                       complex_data=True, npol=2, nchan=1, payload_nbytes=128 << 20,
                                       start_time=np.datetime64('2013-07-02T01:39:20'), telescope='GMRT')
            spf = h0.samples_per_frame
            chunk = torch.view_as_complex((torch.randn(spf, 2, 2, device='cuda', generator=g) * 20.).contiguous())
            opener = lambda: bb.dada.open(os.path.join(tmp, '{utc_start}_{obs_offset:016d}.{file_nr:06d}.dada'), 'ws', header0=h0)
            nchunk = 32
        else:
            h0 = VDIFHeader.fromvalues(edv=0, time=np.datetime64('2014-06-13T05:30:01'), nchan=1, bps=2, complex_data=False,
                                       thread_id=0, samples_per_frame=32000, station='AA')
            chunk = torch.randn(4096 * 32000, device='cuda', generator=g) * 2.
            opener = lambda: bb.vdif.open(os.path.join(tmp, 'f{file_nr:04d}.vdif'), 'ws', header0=h0, sample_rate=32e6,
                                          nthread=1, file_size=8032 * 32768)
            nchunk = 128
        best = None
        for rnd in range(3):
            for n in os.listdir(tmp):
                os.remove(os.path.join(tmp, n))
            torch.cuda.synchronize()
            t = time.perf_counter()
            with opener() as fw:
                for _ in range(nchunk):
                    fw.write(chunk)
            dt = time.perf_counter() - t
            size = sum(os.path.getsize(os.path.join(tmp, n)) for n in os.listdir(tmp))
            nfiles = len(os.listdir(tmp))
            best = dt if best is None else min(best, dt)
        print("RESULT %s %d files, %.2f GB: best of 3 %.3f s = %.2f GB/s of file bytes" % (fmt, nfiles, size / 1e9, best, size / best / 1e9), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
        sys.exit(0)
    for fmt in ('dada', 'vdif'):
        for label, env in (("sink off (BB_WRITE_ASYNC=0)", {"BB_WRITE_ASYNC": "0"}), ("sink, 1 thread", {"BB_WRITE_THREADS": "1"}),
                           ("sink, 2 threads", {"BB_WRITE_THREADS": "2"}), ("sink, 4 threads (default)", {"BB_WRITE_THREADS": "4"}),
                           ("sink, 8 threads", {"BB_WRITE_THREADS": "8"})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', fmt], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=600)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
            print("%-30s %s" % (label, out[0][7:] if out else 'FAILED: ' + (r.stderr or r.stdout)[-300:]), flush=True)
