"""`baseband_amd.empty_output`: an output tensor chosen among several allocations."""
import numpy as np
import pytest

from conftest import golden_path, load_expected, bits_equal

pytestmark = pytest.mark.gpu


def test_empty_output_is_a_plain_tensor_and_decodes_right(manifest):
    import torch
    import baseband_amd
    from baseband_amd import vdif
    case = manifest['vdif_cfg2_small']
    exp = load_expected('vdif_cfg2_small')
    rates = []
    out = baseband_amd.empty_output((exp.shape[0],), candidates=3, report=rates)
    assert out.shape == (exp.shape[0],) and out.dtype == torch.float32 and out.is_cuda
    assert len(rates) == 3 and all(r > 0 for r in rates)
    with vdif.open(golden_path(case['file']), 'rs', sample_rate=case['frame_rate'] * case['samples_per_frame']) as fh:
        assert fh.read(out=out) is out
    assert bits_equal(out.cpu().numpy(), exp.reshape(-1))
    c = baseband_amd.empty_output((1000, 8, 16), dtype=torch.complex64, candidates=2)
    assert c.shape == (1000, 8, 16) and c.dtype == torch.complex64
    one = baseband_amd.empty_output((64000,), candidates=1)
    assert one.numel() == 64000
    with pytest.raises(ValueError):
        from baseband_amd.placement import probe_rate
        probe_rate(torch.empty(100, device='cuda'))
