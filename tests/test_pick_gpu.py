"""k_decode_pick (csrc/k_pick.h, round 5): folded channel subsets with one work
item per wave and direct-to-LDS 16-byte loads.  Bit-exact against a host
expansion from the ORACLE's level tables (`bb_oracle_np.code_levels`, pinned to the
reference by tests/golden/levels.json -- not the table of the library under
test) AND against k_decode_gather_select on the
same input, for every sample width, real / complex chunks, 1-32 thread slots,
payloads at every 4-byte alignment and at odd addresses, ragged last items,
missing frames, offsets outside the buffer, looping workgroups, staged sizes."""
import numpy as np
import pytest

import bb_oracle_np as orc

pytestmark = pytest.mark.gpu

CODERS = {'vdif': 0, 'int': 2}


def _expand(raw, src, nsets, nslot, pn, lev, bps, chunk, within, cplx, fill, int8=False):
    """(nsets * R, nslot, nsel) float32 the kernel must produce."""
    per = 8 // bps
    R = pn * per // chunk
    out = np.empty((nsets, R, nslot, len(within)), np.float32)
    sh = np.arange(0, 8, bps, dtype=np.uint8)
    fre, fim = np.float32(complex(fill).real), np.float32(complex(fill).imag)
    for f in range(nsets):
        for s in range(nslot):
            so = int(src[f * nslot + s])
            if so < 0 or so + pn > raw.size:
                for k, w in enumerate(within):
                    out[f, :, s, k] = fim if (cplx and (w & 1)) else fre
                continue
            b = raw[so:so + pn]
            codes = ((b[:, None] >> sh) & ((1 << bps) - 1)).reshape(R, chunk)
            vals = codes.astype(np.int8).astype(np.float32) if int8 else lev[codes]
            out[f, :, s, :] = vals[:, within]
    return out.reshape(-1)


@pytest.mark.parametrize('bps,chunk,cplx,coder', [(2, 32, True, 'vdif'), (2, 16, False, 'vdif'), (1, 64, False, 'vdif'),
                                                   (4, 8, True, 'vdif'), (8, 4, True, 'vdif'), (8, 8, False, 'int'),
                                                   (2, 4, True, 'vdif')])
@pytest.mark.parametrize('nslot', [1, 2, 8, 32])
def test_pick_matches_host_expansion_and_the_gather_kernel(bps, chunk, cplx, coder, nslot):
    import torch
    from baseband_amd import kernels, _lib
    rng = np.random.default_rng(bps * 1000 + chunk * 10 + nslot)
    lev = orc.code_levels(coder, bps)
    nsets = 7
    try:
        for pn, hdr, lead in ((8000, 32, 0), (5000, 32, 4), (1032, 16, 8), (2000, 20, 12), (1000, 33, 1), (264, 32, 0)):
            if (pn * 8 // bps) % chunk:
                continue
            stride = hdr + pn
            raw = rng.integers(0, 256, lead + nsets * nslot * stride + 64, dtype=np.uint8)
            order = rng.permutation(nsets * nslot)
            src = (lead + order * stride + hdr).astype(np.int64)
            src[3] = -1                                         # a missing frame
            src[-2] = raw.size - pn + 5                         # would run over the end of the buffer
            if nslot > 1:
                src[nslot + 1] = -1
            nsel_opts = [n for n in (1, 2, 4, 8) if n <= chunk and (nslot * n) & (nslot * n - 1) == 0 and 4 <= nslot * n <= 256]
            for nsel in nsel_opts:
                if cplx and nsel >= 2:
                    ch = np.sort(rng.choice(chunk // 2, nsel // 2, replace=False))
                    within = np.stack([2 * ch, 2 * ch + 1], 1).reshape(-1).astype(np.int32)
                else:
                    within = np.sort(rng.choice(chunk, nsel, replace=False)).astype(np.int32)
                exp = _expand(raw, src, nsets, nslot, pn, lev, bps, chunk, within, cplx, -2.5 + 1.5j if cplx else -2.5,
                              int8=coder == 'int')
                d = kernels.to_device_bytes(raw)
                dsrc = torch.from_numpy(src).cuda()
                dw = torch.from_numpy(within).cuda()
                fill = (-2.5 + 1.5j) if cplx else -2.5
                for blocks, pick_bytes in ((0, 8192), (3, 8192), (0, 1024), (0, 32768)):
                    kernels.tune(_lib.TUNE_SELECT_PICK, 2)             # wherever its conditions hold
                    kernels.tune(_lib.TUNE_BLOCKS, blocks)
                    kernels.tune(_lib.TUNE_PICK_BYTES, pick_bytes)
                    got = kernels.decode_frames(d, nsets, pn, CODERS[coder], bps, chunk=chunk, nslot=nslot, src=dsrc,
                                                complex_data=cplx, fill_value=fill, within=dw)
                    name = _lib.last_kernel()
                    assert np.array_equal(got.cpu().numpy().view(np.uint32), exp.view(np.uint32)), \
                        (pn, hdr, lead, nsel, blocks, pick_bytes, name)
                    if (bps * chunk) % 8 == 0 and pick_bytes // nslot >= 256:
                        assert 'k_decode_pick' in name, name
                kernels.tune(_lib.TUNE_SELECT_PICK, 0)
                kernels.tune(_lib.TUNE_BLOCKS, 0)
                old = kernels.decode_frames(d, nsets, pn, CODERS[coder], bps, chunk=chunk, nslot=nslot, src=dsrc,
                                            complex_data=cplx, fill_value=fill, within=dw)
                assert 'k_decode_gather_select' in _lib.last_kernel()
                assert torch.equal(old.view(torch.int32), got.view(torch.int32))
    finally:
        kernels.tune(_lib.TUNE_SELECT_PICK, 1)
        kernels.tune(_lib.TUNE_BLOCKS, 0)
        kernels.tune(_lib.TUNE_PICK_BYTES, 4096)


def test_pick_is_what_a_reader_subset_launches():
    """The cfg3 shape through the API: 8 threads x 16 complex channels, subset of
    2 channels -> k_decode_pick, same samples as decode-then-index."""
    import io
    import torch
    from baseband_amd import vdif, synth, _lib
    image, h0 = synth.random_vdif(11, 50, nthread=8, nchan=16, bps=2, complex_data=True, payload_nbytes=8000,
                                  frame_rate=100, thread_order=[1, 3, 5, 7, 0, 2, 4, 6], invalid=[(4, 2)])
    rate = 100 * h0.samples_per_frame
    with vdif.open(io.BytesIO(image.tobytes()), 'rs', sample_rate=rate, squeeze=False) as fh:
        whole = fh.read()
    with vdif.open(io.BytesIO(image.tobytes()), 'rs', sample_rate=rate, squeeze=False,
                   subset=(slice(None), [3, 12])) as fh:
        part = fh.read()
        assert 'k_decode_pick' in _lib.last_kernel(), _lib.last_kernel()
    assert torch.equal(torch.view_as_real(part).view(torch.int32),
                       torch.view_as_real(whole[:, :, [3, 12]].contiguous()).view(torch.int32))
