#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the small, tracked
summaries under profiles/ and refresh profiles/traffic_latest.json.

usage: tools/summarize_prof.py <round tag> <stats dir> <fetch dir> <write dir> [headline-only stats dir]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return files[0]


def short(name):
    return name if len(name) <= 120 else name[:117] + '...'


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    out_dir = os.path.join(ROOT, 'profiles')
    rows = list(csv.DictReader(open(one(stats_dir + '/**/*kernel_stats.csv'))))
    with open(os.path.join(out_dir, tag + '_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'],
                        r['Percentage'], r['MinNs'], r['MaxNs']])
    res = {}
    for key, d in (('FETCH_SIZE', fetch_dir), ('WRITE_SIZE', write_dir)):
        got = []
        for r in csv.DictReader(open(one(d + '/**/*counter_collection.csv'))):
            if 'k_decode' in r['Kernel_Name'] and r['Counter_Name'] == key:
                got.append((int(float(r.get('Grid_Size') or 0)), float(r['Counter_Value'])))
        # the headline launches: the largest grid (the output arena probes new
        # memory with short launches of the same kernel template)
        top = max(g for g, _ in got)
        res[key] = [v for g, v in got if g == top]
    with open(os.path.join(out_dir, tag + '_pmc_decode.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'counter', 'launch', 'value_KiB'])
        for key, vals in res.items():
            for i, v in enumerate(vals):
                w.writerow(['k_decode_flat', key, i, v])
    dec = [r for r in rows if 'k_decode' in r['Name']]
    if len(sys.argv) > 5:
        # a pass over the headline leg alone: its decode row is not mixed with
        # the other legs' launches of the same kernel
        head = list(csv.DictReader(open(one(sys.argv[5] + '/**/*kernel_stats.csv'))))
        with open(os.path.join(out_dir, tag + '_kernel_stats_headline.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
            for r in head:
                w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'],
                            r['Percentage'], r['MinNs'], r['MaxNs']])
        dec = [r for r in head if 'k_decode' in r['Name']] or dec
    fetch = sum(res['FETCH_SIZE']) / len(res['FETCH_SIZE']) * 1024
    write = sum(res['WRITE_SIZE']) / len(res['WRITE_SIZE']) * 1024
    traffic = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                  "profiles/%s_pmc_decode.csv" % tag,
        "fetch_bytes_raw": fetch, "write_bytes": write,
        "fetch_bytes_corrected": 2 * fetch,
        "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
        "hbm_bytes_per_launch": 2 * fetch + write,
        "kernel_avg_ns_rocprof": float(dec[0]['AverageNs']) if dec else None,
    }
    with open(os.path.join(out_dir, 'traffic_latest.json'), 'w') as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == '__main__':
    main()
