"""Soak of the output arena under threads: four threads open / read / close readers
of a 72 MB file (1.15 GB decoded: arena blocks, `bb_arena_prepare` from every open)
while the idle watcher trims after 50 ms of nothing alive (BB_ARENA_IDLE_S=0.05):
growths, background growths, frees, trims and regrows interleave.  Every read is
checked against the direct decode's checksum.  (a trim must never unmap
under a kernel; VERDICT r4 next 3: prepare / alloc / trim from several threads.)"""
import os
import sys
import threading
import time

os.environ.setdefault('BB_ARENA_IDLE_S', '0.05')
os.environ.setdefault('BB_ARENA_STEP_GIB', '4')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baseband_amd import vdif, synth, kernels, _lib, arena, placement     # noqa: E402

kernels.init()
image, h0 = synth.random_vdif(5, 9000, payload_nbytes=8000, frame_rate=1000)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_stress_arena.vdif')
image.tofile(path)
dev = torch.from_numpy(image.copy()).cuda()
want = float(kernels.decode_frames(dev, 9000, 8000, _lib.CODER_VDIF, 2, src0=32, src_stride=8032).double().sum())
errors, done = [], [0]


def worker(tid):
    try:
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for k in range(40):
                placement._idle_trims = 0           # (keep the delay at 50 ms: this soak wants many trims)
                with vdif.open(path, 'rs', sample_rate=32e6) as fh:
                    got = fh.read()
                    s = float(got.double().sum())
                    if s != want:
                        errors.append((tid, k, s))
                    del got
                done[0] += 1
                if k % 4 == tid or k % 7 == 0:
                    time.sleep(0.09)            # long enough for the idle trim when the others pause too
    except Exception as exc:            # noqa: BLE001
        errors.append((tid, repr(exc)))


ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
t0 = time.time()
[t.start() for t in ths]
[t.join() for t in ths]
ar = arena.default()
st = ar.stats() if ar is not None else {}
print("reads %d in %.1f s, errors %s; arena: grown %.0f GiB, trimmed %.0f GiB, prepares %d, va ranges %d, idle trims %d"
      % (done[0], time.time() - t0, errors[:3], st.get('bytes_grown', 0) / 2 ** 30, st.get('bytes_trimmed', 0) / 2 ** 30,
         st.get('prepares', 0), st.get('va_ranges', 0), placement._idle_trims))
os.remove(path)
sys.exit(1 if errors else 0)
