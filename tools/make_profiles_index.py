"""Writes profiles/INDEX.md: every file under profiles/, grouped by round and
kind, with the places (sources, headers, tests) that cite it.
    python tools/make_profiles_index.py [--archive-uncited <round>]

``--archive-uncited 05``: files of rounds BEFORE round 05 that nothing cites
are moved into ``profiles/archive_uncited_before_r05.tar.gz`` (VERDICT r4 next
8: 176 of 372 files were cited by nothing) -- still in the tree, one file, and
listed by name in the index."""
import os
import re
import subprocess
import sys
import tarfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')
files = sorted(f for f in os.listdir(P) if f != 'INDEX.md')
# who cites what
cited = {}
srcs = subprocess.run(['git', '-C', ROOT, 'ls-files'], capture_output=True, text=True).stdout.split()
srcs = [s for s in srcs if not s.startswith('profiles/') and s.rsplit('.', 1)[-1] in
        ('md', 'py', 'h', 'hip', 'inc', 'c', 'cpp', 'sh', 'json')]
texts = {}
for s in srcs:
    try:
        with open(os.path.join(ROOT, s), errors='replace') as f:
            texts[s] = f.read()
    except OSError:
        pass
import fnmatch
# citations with wildcards (`r04d_exp_glds3_box*.log`, `r05b/c_grow_probe*.log`, `r05k_*`)
patterns = {}
for src, t in texts.items():
    for tok in set(re.findall(r'r\d\d[A-Za-z0-9_./*-]*\*[A-Za-z0-9_.*-]*', t)):
        if len(tok.split('*')[0]) >= 5:                 # (`r03*` in a sentence about a whole round is not a citation)
            patterns.setdefault(tok.rstrip('.'), set()).add(src)
for f in files:
    stem = f.rsplit('.', 1)[0]
    who = [s for s, t in texts.items() if f in t or (stem + '*') in t or (stem.rsplit('_', 1)[0] + '*') in t]
    for pat, srcs in patterns.items():
        if fnmatch.fnmatch(f, pat) or fnmatch.fnmatch(f, pat + '*'):
            who.extend(srcs)
    cited[f] = sorted(set(who))


archived = []
if '--archive-uncited' in sys.argv:
    cur = sys.argv[sys.argv.index('--archive-uncited') + 1]
    tar_name = 'archive_uncited_before_r{}.tar.gz'.format(cur)
    victims = [f for f in files if not cited[f] and re.match(r'r(\d\d)', f) and re.match(r'r(\d\d)', f).group(1) < cur
               and not f.startswith('archive_')]
    if victims:
        old = []
        tar_path = os.path.join(P, tar_name)
        if os.path.exists(tar_path):                 # keep what an earlier run put there
            import io
            with tarfile.open(tar_path, 'r:gz') as t:
                old = [(m, t.extractfile(m).read() if m.isfile() else None) for m in t.getmembers()]
        with tarfile.open(tar_path, 'w:gz') as t:
            for m, data in old:
                if data is not None:
                    t.addfile(m, io.BytesIO(data))
            for f in victims:
                t.add(os.path.join(P, f), arcname=f)
        import shutil
        for f in victims:
            path = os.path.join(P, f)
            shutil.rmtree(path) if os.path.isdir(path) else os.remove(path)
        archived = victims
        files = [f for f in files if f not in victims]
        if tar_name not in files:
            files.append(tar_name)
            cited[tar_name] = ['tools/make_profiles_index.py']
        files.sort()
for f in files:
    if f.startswith('archive_uncited'):
        cited.setdefault(f, ['tools/make_profiles_index.py'])


def kind(f):
    if 'kernel_stats' in f or f.endswith('_kernels.csv'):
        return 'rocprofv3 --kernel-trace --stats summaries'
    if '_pmc_' in f or 'traffic' in f:
        return 'rocprofv3 --pmc counter passes (HBM traffic)'
    if re.search(r'_bench(_plain|_under_rocprof)?\.json$', f) or re.search(r'_bench\.json$', f) or '/bench' in f:
        return 'bench.py lines'
    if '_exp_' in f or 'arena_probe' in f or '_dbg_' in f:
        return 'A/B experiments (tools/experiments/)'
    if '_bench_' in f or '_prof_' in f or '_kbench' in f or 'locate' in f or 'tfpick' in f or 'pipeline' in f:
        return 'benchmarks and host profiles (tools/)'
    return 'other records'


rounds = {}
for f in files:
    m = re.match(r'r(\d\d)', f)
    rounds.setdefault(m.group(1) if m else 'zz', []).append(f)
out = ["# profiles/ -- index", "",
       "Every measurement file of rounds 1-6, grouped by round and kind.  `rNN<x>_` = round NN, run x (a, b, ... z, za, ...).",
       "\"cited in\" lists the documents and sources whose statements rest on the file.  Regenerate with",
       "`python tools/make_profiles_index.py`.", "",
       "Where to start: the newest `*_bench.json` (the driver-format line), the `*_kernel_stats*.csv` of the same run",
       "(rocprofv3 per-kernel averages: must agree with the line's `roofline.kernel_ms_avg`), `traffic_latest.json` /",
       "`*_pmc_decode.csv` (HBM bytes per launch), then DESIGN.md, which cites the experiment logs by name.", ""]
for f in files:
    if f.startswith('archive_uncited') and f.endswith('.tar.gz'):
        with tarfile.open(os.path.join(P, f), 'r:gz') as t:
            inside = sorted({m.name.split('/')[0] for m in t.getmembers()})
        out += ["## Archived: `{}` ({} entries nothing cites)".format(f, len(inside)), "",
                ', '.join('`{}`'.format(n) for n in inside), ""]
names = {'01': 'Round 1', '02': 'Round 2', '03': 'Round 3', '04': 'Round 4', '05': 'Round 5', 'zz': 'Not tied to a round'}
for r in sorted(rounds, reverse=True):
    out.append("## {} ({} files)".format(names.get(r, 'Round ' + r), len(rounds[r])))
    out.append("")
    by = {}
    for f in rounds[r]:
        by.setdefault(kind(f), []).append(f)
    for k in sorted(by):
        out.append("### " + k)
        out.append("")
        out.append("| file | cited in |")
        out.append("|---|---|")
        for f in by[k]:
            path = os.path.join(P, f)
            label = f + ('/' if os.path.isdir(path) else '')
            out.append("| `{}` | {} |".format(label, ', '.join('`{}`'.format(c) for c in cited[f][:6]) or '--'))
        out.append("")
with open(os.path.join(P, 'INDEX.md'), 'w') as f:
    f.write('\n'.join(out) + '\n')
print(len(files), "files indexed")
