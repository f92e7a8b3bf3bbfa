"""Timeline of mid-size reads from a rocprofv3 run of tools/prof_read_trace.py:
    python tools/summarize_read_trace.py OUTDIR [marks file]
Per read (the last ten): host time from read() to the first kernel launch call, from there
to the decode's launch call, launch call -> kernel start for scan / index / decode, their
durations, and decode end -> host's synchronize returns.  All from the kernel trace (start /
end of every dispatch) and the HIP runtime trace (hipLaunchKernel / hipModuleLaunchKernel calls)."""
import csv
import glob
import os
import sys

import numpy as np

out = sys.argv[1]
marks = [tuple(int(x) for x in ln.split()) for ln in open(sys.argv[2] if len(sys.argv) > 2 else 'read_marks.txt')]


def load(pattern):
    files = glob.glob(os.path.join(out, '**', pattern), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


kern = load('*kernel_trace.csv')
api = load('*hip_api_trace.csv')
k = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Correlation_Id', 0))) for r in kern))
launches = {int(r['Correlation_Id']): (int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function'])
            for r in api if 'Launch' in r['Function']}
# clock of the marks (time.time_ns, CLOCK_REALTIME) against the trace's clock: the first launch
# call after a mark cannot come before it -- take the smallest such distance over all reads as offset 0
apis = sorted((int(r['Start_Timestamp']), r['Function']) for r in api)
rows = []
for t0, t1, t2 in marks[4:]:
    pass
print('kernels', len(k), 'launch calls', len(launches), 'reads', len(marks))
# group kernels into reads by the decode kernel: every k_decode_flat_lds dispatch with a big grid closes one read
dec = [x for x in k if 'k_decode_flat_lds' in x[2]]
scan = [x for x in k if x[2].startswith('k_vdif_scan')]
idx = [x for x in k if x[2].startswith('k_index_verify')]
n = min(len(dec), len(scan), len(idx), 10)
table = []
for d, s_, i_ in list(zip(dec[-n:], scan[-n:], idx[-n:])):
    ls, li, ld = launches.get(s_[3]), launches.get(i_[3]), launches.get(d[3])
    if not (ls and li and ld):
        continue
    # the host call before the scan's launch that starts the read: the last hipStreamSynchronize /
    # hipDeviceSynchronize END before it (the loop syncs before every read)
    sync_end = max((int(r['End_Timestamp']) for r in api if 'Synchronize' in r['Function']
                    and int(r['End_Timestamp']) <= ls[0]), default=ls[0])
    sync_after = min((int(r['End_Timestamp']) for r in api if 'Synchronize' in r['Function']
                      and int(r['End_Timestamp']) >= d[1]), default=d[1])
    table.append(dict(host_to_scan_launch=(ls[0] - sync_end) / 1e3, scan_launch_to_start=(s_[0] - ls[0]) / 1e3,
                      scan=(s_[1] - s_[0]) / 1e3, scan_end_to_index_start=(i_[0] - s_[1]) / 1e3, index=(i_[1] - i_[0]) / 1e3,
                      decode_launch_call_after_scan_launch=(ld[0] - ls[0]) / 1e3,
                      index_end_to_decode_start=(d[0] - i_[1]) / 1e3, decode=(d[1] - d[0]) / 1e3,
                      decode_end_to_sync_return=(sync_after - d[1]) / 1e3,
                      total=(sync_after - sync_end) / 1e3))
for key in table[0]:
    print('%-40s %8.1f us (median of %d reads)' % (key, float(np.median([t[key] for t in table])), len(table)))
