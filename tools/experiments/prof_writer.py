"""Host profile of the VDIF stream writer: 2 GiB of cfg2 frames, 512 MiB of
float32 per write() call (cProfile; profiles/r04zn_prof_writer.log)."""
import os, sys, time, tempfile, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import baseband_amd as bb
from baseband_amd.vdif.header import VDIFHeader

dev = torch.device('cuda', 0)
tmp = tempfile.mkdtemp(prefix='bb_wr_')
path = os.path.join(tmp, 'cfg2.vdif')
g = torch.Generator(device=dev); g.manual_seed(1)
chunk = torch.randn(4096 * 32000, device=dev, generator=g) * 2.
h0 = VDIFHeader.fromvalues(edv=0, time=np.datetime64('2014-06-13T05:30:01'), nchan=1, bps=2, complex_data=False,
                           thread_id=0, samples_per_frame=32000, station='AA')
# reference: what a thread of this process puts into a NEW file from a pinned buffer, 32 MiB pieces
import threading
ref = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
ref.numpy()[:] = 7
for how in ('main thread', 'a second thread'):
    def wr():
        with open(path + '.ref', 'wb', buffering=0) as f:
            for lo in range(0, 1 << 30, 32 << 20):
                f.write(memoryview(ref.numpy()[lo:lo + (32 << 20)]))
    t = time.perf_counter()
    if how == 'main thread':
        wr()
    else:
        th = threading.Thread(target=wr); th.start(); th.join()
    print('reference: 1 GiB pinned -> new file by write() of 32 MiB pieces on %s: %.2f GB/s' % (how, (1 << 30) / (time.perf_counter() - t) / 1e9), flush=True)
    os.remove(path + '.ref')
from baseband_amd import staging
for mode in ('async', 'sync'):
  staging._WRITE_ASYNC = mode == 'async'
  print('## BB_WRITE_ASYNC', mode)
  for rnd in range(3):
      pr = cProfile.Profile() if rnd == 2 else None
      torch.cuda.synchronize()
      t = time.perf_counter()
      if pr: pr.enable()
      with bb.vdif.open(path, 'ws', header0=h0, sample_rate=32e6, nthread=1) as fw:
          for _ in range(65):
              fw.write(chunk)
      if pr: pr.disable()
      dt = time.perf_counter() - t
      print('round', rnd, 'write %.3f s = %.2f GB/s of file bytes' % (dt, os.path.getsize(path) / dt / 1e9), flush=True)
      os.remove(path)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:4500])
