"""Host profile of random access on a file: seek to a random frame, read one
frame's worth of samples (page cache -> HBM window -> scan / index / decode)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth   # noqa: E402

tmp = os.environ.get('TMPDIR', '/tmp')
path = os.path.join(tmp, 'bb_rr.vdif')
image, h0 = synth.random_vdif(1, (1 << 28) // 8032, payload_nbytes=8000, frame_rate=1000)
image.tofile(path); del image
rng = np.random.default_rng(0)
for verify, resident in ((True, False), (True, True)):
    with vdif.open(path, 'rs', sample_rate=32e6, verify=verify) as fh:
        if resident:
            fh.stage()
        nfr = fh.shape[0] // 32000
        where = rng.integers(0, nfr - 2, 600)
        for k in where[:100]:
            fh.seek(int(k) * 32000 + 137); d = fh.read(32000)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t = time.perf_counter()
        pr.enable()
        for k in where[100:]:
            fh.seek(int(k) * 32000 + 137)
            d = fh.read(32000)
        pr.disable()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("resident" if resident else "from the page cache", "us per read (under cProfile):", round(dt / 500 * 1e6, 1))
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
        out = s.getvalue()
        print(out[out.index('ncalls'):][:3600])
os.remove(path)
