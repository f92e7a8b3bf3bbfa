"""Ten fh.read(2^15 frames) with a host sync between them, for a rocprofv3 timeline:
    rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d OUT -- python3 tools/prof_read_trace.py
tools/summarize_read_trace.py OUT turns the two traces into: host time before the first
launch of a read, launch -> start gaps, scan / index / decode durations, end of decode ->
return of the host's synchronize."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from baseband_amd import vdif, kernels                  # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (1 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
SPF = bench.SPF
nf = 1 << 15
marks = []
with vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) as fh:
    for k in range(14):
        fh.seek(((k * 7 + 2) * nf % (nframes - nf)) * SPF)
        torch.cuda.synchronize()
        t0 = time.time_ns()
        got = fh.read(nf * SPF)
        t1 = time.time_ns()
        torch.cuda.synchronize()
        t2 = time.time_ns()
        marks.append((t0, t1, t2))
        del got
out = os.environ.get('BB_TRACE_MARKS', 'read_marks.txt')
with open(out, 'w') as f:
    for m in marks:
        f.write('%d %d %d\n' % m)
print('reads: returns after %.1f us, done after %.1f us (medians of the last 10)' % (
    sorted((b - a) / 1e3 for a, b, c in marks[4:])[5], sorted((c - a) / 1e3 for a, b, c in marks[4:])[5]))
