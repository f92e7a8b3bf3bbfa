#!/usr/bin/env python3
"""Condense tools/prof_kernels.sh output into profiles/<tag>_kernels.csv: one
row per decode kernel with average duration, HBM bytes from the counters
(FETCH_SIZE x2 on gfx950) and the SQ / LDS counters.
usage: summarize_kernels.py <tag> <dir>"""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], sys.argv[2]


def one(pat):
    f = glob.glob(pat, recursive=True)
    return f[0] if f else None


def short(n):
    n = n.split('(')[0]
    return n.replace('void ', '')


stats = {}
f = one(d + '/stats/**/*kernel_stats.csv')
if f:
    for r in csv.DictReader(open(f)):
        if 'k_decode' in r['Name']:
            stats[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']), float(r['MinNs']), float(r['MaxNs']))
ctr = defaultdict(lambda: defaultdict(list))
for name in ('fetch', 'write', 'sq'):
    f = one(d + '/%s/**/*counter_collection.csv' % name)
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if 'k_decode' in r['Kernel_Name']:
            ctr[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
cols = ['FETCH_SIZE', 'WRITE_SIZE', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
        'SQ_ACTIVE_INST_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS']
outp = os.path.join(ROOT, 'profiles', tag + '_kernels.csv')
with open(outp, 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'calls', 'avg_ms', 'min_ms', 'max_ms', 'hbm_read_GB(FETCH_SIZEx2)', 'hbm_write_GB',
                'hbm_TBps'] + cols[2:])
    for k in sorted(set(stats) | set(ctr)):
        calls, avg, mn, mx = stats.get(k, (0, 0., 0., 0.))
        c = ctr.get(k, {})
        mean = lambda key: (sum(c[key]) / len(c[key])) if c.get(key) else None
        rd = mean('FETCH_SIZE'); wr = mean('WRITE_SIZE')
        rd = None if rd is None else 2 * rd * 1024 / 1e9
        wr = None if wr is None else wr * 1024 / 1e9
        tb = None if (rd is None or wr is None or not avg) else (rd + wr) / (avg * 1e-9) / 1e3
        w.writerow([k, calls, round(avg / 1e6, 4), round(mn / 1e6, 4), round(mx / 1e6, 4),
                    None if rd is None else round(rd, 3), None if wr is None else round(wr, 3),
                    None if tb is None else round(tb, 3)] + [None if mean(x) is None else int(mean(x)) for x in cols[2:]])
print(open(outp).read())
