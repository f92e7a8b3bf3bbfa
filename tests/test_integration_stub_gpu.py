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


def test_plugin_modules_return_numpy_arrays(manifest, tmp_path):
    """The ``baseband.io`` entry points (baseband_amd/plugin/): ``read()`` hands a
    caller of the reference the NumPy array the reference would (bit-identical to
    its output), ``read_tensor()`` the device tensor, ``out=`` a NumPy array or a
    device tensor is filled in place."""
    import torch
    from conftest import golden_path, load_expected, bits_equal
    from baseband_amd.plugin import vdif as pv, mark5b as pm, dada as pd
    exp = load_expected('sample_vdif')
    with pv.open(golden_path('samples/sample.vdif'), 'rs') as fh:
        got = fh.read()
        assert isinstance(got, np.ndarray) and bits_equal(got.reshape(exp.shape), exp)
        fh.seek(0)
        t = fh.read_tensor(1000)
        assert isinstance(t, torch.Tensor) and t.is_cuda and bits_equal(t.cpu().numpy().reshape(exp[:1000].shape), exp[:1000])
        out = np.empty((500,) + got.shape[1:], got.dtype)
        assert fh.read(out=out) is out and bits_equal(out.reshape(exp[1000:1500].shape), exp[1000:1500])
        dev = torch.empty((250,) + got.shape[1:], dtype=torch.float32, device='cuda')
        assert fh.read(out=dev) is dev and bits_equal(dev.cpu().numpy().reshape(exp[1500:1750].shape), exp[1500:1750])
    c = manifest['m5b_c16_b2']
    with pm.open(golden_path(c['file']), 'rs', sample_rate=c['frame_rate'] * c['samples_per_frame'], kday=c['kday'],
                 nchan=c['nchan'], bps=c['bps']) as fh:
        assert bits_equal(fh.read(), load_expected('m5b_c16_b2'))
    with pd.open(golden_path(manifest['dada_p2_c4_cplx']['file']), 'rs') as fh:
        got = fh.read()
        assert got.dtype == np.complex64 and bits_equal(got, load_expected('dada_p2_c4_cplx'))
    # a writer through the plugin module takes NumPy samples
    with pv.open(golden_path('samples/sample.vdif'), 'rs') as fr:
        h0, data = fr.header0, fr.read()
    with pv.open(str(tmp_path / 'again.vdif'), 'ws', header0=h0, sample_rate=32e6, nthread=8) as fw:
        fw.write(data)
    # (the writer stores the threads in increasing order, sample.vdif holds them as 1, 3, 5, 7, 0, 2, 4, 6:
    # the same samples, not the same bytes; byte identity of the writers is tests/test_encode_gpu.py's)
    with pv.open(str(tmp_path / 'again.vdif'), 'rs') as fr:
        assert fr.header0['thread_id'] == 0 and fr.shape == data.shape
        assert bits_equal(fr.read(), data)


def test_plugin_binary_readers_hand_out_numpy_frames(tmp_path):
    """'rb' / 'wb' through the plugin modules: ``read_frame()`` / ``read_frameset()``
    come back as views whose ``data`` and ``frame[item]`` are NumPy arrays equal to
    the reference's (its frame API: vdif/frame.py:31-223, base/frame.py:160-199),
    headers as header views; a frame read this way can be written back."""
    from conftest import golden_path, load_expected, bits_equal
    from baseband_amd.plugin import vdif as pv, mark5b as pm
    from baseband_amd.plugin._proxy import FrameView, HeaderView, PayloadView
    exp = load_expected('sample_vdif').reshape(40000, 8)    # thread order 0..7
    with pv.open(golden_path('samples/sample.vdif'), 'rb') as fb:
        frame = fb.read_frame()
        assert isinstance(frame, FrameView) and isinstance(frame.header, HeaderView)
        thread = frame.header['thread_id']
        data = frame.data
        assert isinstance(data, np.ndarray) and data.shape == (20000, 1)
        assert bits_equal(data[:, 0], exp[:20000, thread])
        assert isinstance(frame[10:20], np.ndarray) and bits_equal(frame[10:20], data[10:20])
        assert isinstance(frame.payload, PayloadView) and isinstance(frame.payload.data, np.ndarray)
        assert frame.payload.words.dtype == np.dtype('<u4') and frame.nbytes == 5032 and frame.valid
        fb.seek(0)
        fset = fb.read_frameset()
        assert isinstance(fset, FrameView) and isinstance(fset.data, np.ndarray)
        assert fset.data.shape == (20000, 8, 1) and bits_equal(fset.data[..., 0], exp[:20000])
        assert all(isinstance(f, FrameView) for f in fset.frames) and len(fset.frames) == 8
        # ... and written back through a 'wb' view: the same bytes as in the file
        with pv.open(str(tmp_path / 'one.vdif'), 'wb') as fw:
            fw.write_frame(frame)
        with open(golden_path('samples/sample.vdif'), 'rb') as f:
            assert (tmp_path / 'one.vdif').read_bytes() == f.read(5032)
    with pm.open(golden_path('samples/sample.m5b'), 'rb', kday=56000, nchan=8, bps=2) as fb:
        frame = fb.read_frame()
        assert isinstance(frame.data, np.ndarray) and frame.data.shape == (5000, 8)
        assert bits_equal(frame.data, load_expected('sample_m5b')[:5000])
