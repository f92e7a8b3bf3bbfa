#!/usr/bin/env python3
"""Register and scratch use of every kernel in the product library, from the
device ISA (`make -C baseband_amd/csrc asm` -> build/bbdecode.s).  Prints one
line per kernel; exit status 1 if a kernel spills to scratch or a streaming
decode kernel needs more than 128 VGPRs (fewer than 4 waves per SIMD: the LDS
staged byte-table kernel lost 15 % of its rate at 164 VGPRs, profiles/r03m).
    python tools/check_isa.py [--build]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(path=os.path.join(ROOT, 'build', 'bbdecode.s')):
    with open(path) as f:
        s = f.read()
    out = []
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
        body = m.group(2)

        def field(name):
            return int(re.search(r'\.amdhsa_%s (\d+)' % name, body).group(1))
        out.append(dict(name=m.group(1), vgpr=field('next_free_vgpr'), sgpr=field('next_free_sgpr'),
                        scratch=field('private_segment_fixed_size'), lds=field('group_segment_fixed_size')))
    return out


def demangle(names):
    try:
        r = subprocess.run(['c++filt'] + names, capture_output=True, text=True, timeout=60)
        return r.stdout.splitlines()
    except Exception:
        return names


def main():
    if '--build' in sys.argv:
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'baseband_amd', 'csrc'), 'asm'],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ks = kernels()
    names = demangle([k['name'] for k in ks])
    bad = 0
    for k, n in sorted(zip(ks, names), key=lambda x: x[1]):
        n = re.sub(r'^void ', '', n).replace('(bb_flat_args)', '').replace('(bb_gather_args)', '')
        flag = ''
        if k['scratch']:
            flag, bad = ' <-- SCRATCH', bad + 1
        elif k['vgpr'] > 128 and 'k_decode' in n:
            flag, bad = ' <-- more than 128 VGPRs', bad + 1
        print("{:4d} VGPR {:4d} SGPR {:6d} B LDS {:4d} B scratch  {}{}".format(
            k['vgpr'], k['sgpr'], k['lds'], k['scratch'], n[:110], flag))
    print("{} kernels, {} flagged".format(len(ks), bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
