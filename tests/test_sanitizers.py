"""CPU sanitizer build: the C restatement of the
reference algorithm (oracle/bb_oracle.c) is compiled with AddressSanitizer and
UndefinedBehaviorSanitizer and driven over exact-size buffers by
oracle/san_check.c; any over-read, over-write, misaligned access or signed
overflow aborts the run.  (GPU ASan is not available on this pool; the HIP
kernels are covered by the bit-exact parity tests instead.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which('gcc') is None and shutil.which('cc') is None, reason="no C compiler")
def test_c_oracle_under_asan_ubsan():
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'san'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert 'san_check ok' in r.stdout
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr
