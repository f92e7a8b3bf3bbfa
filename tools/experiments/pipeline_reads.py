"""bench.py's pipeline leg with more reads per format, every read printed:
is the first format (cfg2) slower because it is first?  python pipeline_reads.py [reads]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from baseband_amd import kernels
kernels.init()
from baseband_amd import staging, arena
_run = staging.WindowPipeline.run
def run(self, ranges, process, sink=None):
    ranges = list(ranges)
    ar = arena.default(0)
    import time
    t = time.perf_counter()
    _run(self, ranges, process, sink)
    dt = time.perf_counter() - t
    print('   run: %d windows, first %r, cap %d, sink %s in_arena %s, pinned %s, %.2f ms' % (
        len(ranges), ranges[0], self.cap, None if sink is None else hex(sink.data_ptr()),
        None if sink is None or ar is None else ar.owns(sink), [hex(t.data_ptr()) for t in self._pinned if t is not None], dt * 1e3),
        file=sys.stderr, flush=True)
staging.WindowPipeline.run = run
if os.environ.get('NO_LINK_PROBE'):
    bench.pinned_h2d_rate = lambda device: 57.0
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10
_sum = staging.window_trace_summary
def summary(rows):
    print('   per window enqueue_h2d ms:', [round(r['enqueue_h2d_ms'], 2) for r in rows], file=sys.stderr)
    print('   per window enqueue rest ms:', [round(r['enqueue_ms'] - r['enqueue_h2d_ms'], 2) for r in rows], file=sys.stderr)
    print('   per window host copy ms:', [round(r['host_copy_ms'], 2) for r in rows], file=sys.stderr)
    return _sum(rows)
staging.window_trace_summary = summary
d = bench.leg_pipeline(torch.device('cuda', 0), gib=2.0, reads=reads)
for f in d["formats"]:
    print(f["case"][:28], f.get("file_GBps_best"), f.get("file_GBps_median"), 'enqueue', f["windows"]["host_enqueue_ms"], 'of which h2d', f["windows"]["host_enqueue_h2d_ms"])
    print('   read_call ms:', [p[1] for p in f.get("host_ms_of_each_read", [])])
