"""Where do the milliseconds of one open(path).read() go?  (Round 4: the
pipeline leg of bench.py reads 2 GiB files at 42-50 GB/s with the H2D copies
at 56 GB/s and the host copies at 60-100.)  Host timestamps around open, read,
the window loop (first window start .. last window enqueued), the final
synchronisation and close; the per-window sums as bench.py reports them.
    python tools/prof_read_timeline.py [GiB]"""
import json, os, sys, time, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import vdif, synth, staging, kernels, arena
kernels.init()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
nframes = int(gib * 2 ** 30) // 8032
image, h0 = synth.random_vdif(12345, nframes, payload_nbytes=8000, frame_rate=1000)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_timeline.vdif')
image.tofile(path)
size = image.size
del image
for rep in range(6):
    staging.trace = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fh = vdif.open(path, 'rs', sample_rate=32e6)
    t1 = time.perf_counter()
    out = fh.read()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    fh.close()
    t4 = time.perf_counter()
    rows = staging.trace
    staging.trace = None
    summ = staging.window_trace_summary(rows)
    print(json.dumps({"rep": rep, "file_GBps": round(size / (t4 - t0) / 1e9, 2), "total_ms": round((t4 - t0) * 1e3, 2),
                      "open_ms": round((t1 - t0) * 1e3, 2), "read_call_ms": round((t2 - t1) * 1e3, 2),
                      "before_first_window_ms": round((rows[0]["t_start"] - t1) * 1e3, 2),
                      "window_loop_ms": round((rows[-1]["t_end"] - rows[0]["t_start"]) * 1e3, 2),
                      "after_last_window_ms": round((t2 - rows[-1]["t_end"]) * 1e3, 2),
                      "final_sync_ms": round((t3 - t2) * 1e3, 2), "close_ms": round((t4 - t3) * 1e3, 2),
                      "windows": summ,
                      "first_windows": [{k: round(r[k], 2) for k in ("wait_ms", "host_copy_ms", "enqueue_ms")} | {"MiB": r["bytes"] >> 20} for r in rows[:4]],
                      "last_windows": [{k: round(r[k], 2) for k in ("wait_ms", "host_copy_ms", "enqueue_ms")} | {"MiB": r["bytes"] >> 20} for r in rows[-2:]]}), flush=True)
    del out
os.remove(path)
