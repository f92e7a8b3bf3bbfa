"""INTEGRATION.md's ctypes loader is not prose: the code block of section 1 is
extracted and executed as written (only the library path is pointed at the
in-tree build), and its `make_decoder` is used the way section 2 hooks it into
the reference's `_decoders` dicts -- words in, float32 out -- against the
golden outputs."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu


def _stub_namespace():
    with open(os.path.join(ROOT, 'INTEGRATION.md')) as f:
        text = f.read()
    block = re.search(r"## 1\. Loader.*?```python\n(.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libbbdecode.so")',
                          'C.CDLL(%r)' % os.path.join(ROOT, 'baseband_amd', 'libbbdecode.so'))
    import baseband_amd._lib                    # noqa: F401  (torch's HIP runtime first, as _lib.py explains)
    ns = {}
    exec(compile(block, 'INTEGRATION.md#1', 'exec'), ns)
    return ns


def test_documented_binding_decodes_like_the_reference():
    ns = _stub_namespace()
    # section 2: VDIFPayload._decoders = {bps: make_decoder(0, bps)}; sample.vdif frame 0 = thread 1
    raw = load_file('samples/sample.vdif')
    words = raw[32:5032].view('<u4')
    out = ns['make_decoder'](0, 2)(words)
    exp = load_expected('sample_vdif')                  # (40000, 8, 1)
    assert out.dtype == np.float32 and out.shape == (20000,)
    assert bits_equal(out, np.ascontiguousarray(exp[:20000, 1, 0]))
    assert out[:12].astype(int).tolist() == [1, 1, 1, -3, 1, 1, -3, -3, -3, 3, 3, -1]   # vdif/tests/test_vdif.py:381-382
    # Mark 5B (coder 1) and int8 (coder 2) through the same stub
    m5 = load_file('samples/sample.m5b')
    got = ns['make_decoder'](1, 2)(m5[16:10016].view('<u4')).reshape(-1, 8)
    assert bits_equal(got, np.ascontiguousarray(load_expected('sample_m5b')[:5000]))
    i8 = np.arange(-128, 128, dtype=np.int8).view(np.uint8)
    assert ns['make_decoder'](2, 8)(i8.view('<u4')).tolist() == list(range(-128, 128))
    # an unknown coder / bps surfaces as KeyError, like a missing _decoders entry
    with pytest.raises(KeyError):
        ns['make_decoder'](1, 4)(words)
