"""Behaviours the reference's own test-suite checks that no other test here covered
(vdif/tests/test_vdif.py: test_incomplete_stream, test_corrupt_stream,
test_invalid_last_frame, test_io_invalid, test_count_not_changed,
test_find_header_lost_start, test_edf3_vdif_payload_size, test_one_frame_per_second),
restated against this package's API (plain Hz, numpy.datetime64, device tensors)."""
import io
import os
import warnings

import numpy as np
import pytest

from conftest import golden_path

from baseband_amd import vdif
from baseband_amd.base.base import HeaderNotFoundError

pytestmark = pytest.mark.gpu
SAMPLE = golden_path('samples/sample.vdif')


@pytest.mark.parametrize('fill_value', [0., -999.])
def test_incomplete_stream(tmp_path, fill_value):
    p = str(tmp_path / 'incomplete.vdif')
    with vdif.open(SAMPLE, 'rs') as fr:
        record = fr.read(20010)
        with pytest.warns(UserWarning, match='partial buffer'):
            with vdif.open(p, 'ws', header0=fr.header0, sample_rate=32e6, nthread=8) as fw:
                fw.write(record)
    with vdif.open(p, 'rs', fill_value=fill_value) as fwr:
        valid = fwr.read(20000)
        assert bool((valid == record[:20000]).all())
        assert fwr.fill_value == fill_value
        invalid = fwr.read()
        assert invalid.shape == valid.shape
        assert bool((invalid == fill_value).all())


def test_corrupt_stream(tmp_path):
    p = str(tmp_path / 'test.vdif')
    with vdif.open(SAMPLE, 'rb') as fh, open(p, 'w+b') as s:
        frameset = fh.read_frameset()
        header0 = frameset.frames[0].header.copy()
        frameset.tofile(s)
        for frame in frameset.frames:
            frame.header.mutable = True
        for i in range(5):
            frameset['frame_nr'] += 1
            frameset.tofile(s)
        # lots of the final frame, i.e., with the wrong thread_id
        fh.seek(-5032, 2)
        frame2 = fh.read_frame()
        for i in range(15):
            frame2.tofile(s)
        s.seek(0)
        with vdif.open(s, 'rs') as f2:
            assert f2.header0 == header0
            with pytest.raises(HeaderNotFoundError):
                f2._last_header


def test_invalid_last_frame(tmp_path):
    p = str(tmp_path / 'test2.vdif')
    with vdif.open(SAMPLE, 'rs') as fh, open(p, 'ab') as s, \
            vdif.open(s, 'ws', header0=fh.header0, nthread=8) as fw:
        data = fh.read()
        for _ in range(5):
            fw.write(data)
        fw.flush()
        assert s.seek(0, 2) == 5032 * 8 * 2 * 5
        bad_header = vdif.VDIFHeader.fromvalues(edv=0, frame_nbytes=5032, invalid_data=True)
        bad_header.tofile(s)
        s.write(b'\0' * 5000)
        assert s.tell() == 5032 * (8 * 2 * 5 + 1)
        start, stop, n = fh.start_time, fh.stop_time, fh.shape[0]
    with vdif.open(p, 'rs', sample_rate=32e6) as f2:
        assert f2.start_time == start
        assert f2.shape[0] == 5 * n
        assert abs((f2.stop_time - stop) - 4 * (stop - start)) < np.timedelta64(1, 'ns')
        d2 = f2.read()
    assert bool((d2.reshape(5, -1, 8) == data).all())


def test_io_invalid(tmp_path):
    p = str(tmp_path / 'ts.dat')
    with open(p, 'wb') as fw:
        fw.write(b'      ')
    with pytest.raises(TypeError):
        vdif.open(p, 'rb', bla=10)              # extra argument
    with pytest.raises(ValueError):
        vdif.open(p, 's')                       # missing w or r


def test_count_not_changed():
    count = np.array(2)
    with vdif.open(golden_path('samples/sample_arochime.vdif'), 'rs', sample_rate=800e6 / 2048) as fh:
        got = fh.read(count)
        assert count == 2 and got.shape[0] == 2


@pytest.mark.parametrize('name', ['sample.vdif', 'sample_mwa.vdif', 'sample_arochime.vdif', 'sample_bps1.vdif',
                                  'sample_vlbi.vdif'])
def test_find_header_lost_start(name, tmp_path):
    p = str(tmp_path / 'corrupted.vdif')
    with vdif.open(golden_path('samples/' + name), 'rb') as fh, open(p, 'wb') as fw:
        h0 = fh.read_header()
        fh.seek(h0.frame_nbytes)
        h1 = fh.read_header()
        fh.seek(h0.frame_nbytes - 100)
        fw.write(fh.read())
    with vdif.open(p, 'rb') as fc:
        header = fc.find_header()
        assert header is not None
        assert header == h1


def test_edv3_vdif_payload_size(tmp_path):
    p = str(tmp_path / 'test.vdif')
    with vdif.open(SAMPLE, 'rs') as fh:
        header1 = fh.header0.copy()
        header1.payload_nbytes = 1000
        data1 = fh.read()
        with vdif.open(p, 'ws', header0=header1, nthread=8) as fw:
            fw.write(data1)
    with vdif.open(p, 'rs') as fc:
        header2 = fc.header0
        data2 = fc.read()
        assert header2.frame_nbytes == 1032
        assert header2.payload_nbytes == 1000
        assert header2.samples_per_frame == 4000
        assert bool((data2 == data1).all())


def test_one_frame_per_second(tmp_path):
    p = str(tmp_path / 'test.vdif')
    with vdif.open(SAMPLE, 'rs') as fh:
        header1 = fh.header0.copy()
        header1.frame_rate = 1.
        data1 = fh.read()
        with vdif.open(p, 'ws', header0=header1, nthread=8) as fw:
            fw.write(data1)
            stop_time = fw.time
    with vdif.open(p, 'rs') as fc:
        assert fc._frame_rate == 1
        assert abs(fc.stop_time - stop_time) < np.timedelta64(1, 'ns')
        data2 = fc.read()
        assert bool((data2 == data1).all())
