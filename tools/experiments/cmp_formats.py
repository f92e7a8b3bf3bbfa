"""Table of tools/bench_formats.py lines that carry GBps_by_stripes (BB_WORK_STRIPES_SWEEP)."""
import json
import sys

for f in sys.argv[1:]:
    print('==', f)
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith('{'):
            continue
        d = json.loads(ln)
        by = d.get('GBps_by_stripes')
        if not by:
            continue
        base = d['algorithmic_GBps']
        print("%-66s product %6.0f | %s" % (d['case'][:66], base, "  ".join(
            "%s: %6.0f (%+.1f%%)" % (k, v, (v / base - 1) * 100) for k, v in by.items())))
