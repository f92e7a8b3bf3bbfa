"""Sample shapes under squeeze and subset, for every format with more than one sample axis."""
from ._dsl import *    # noqa: F401,F403

ARO = dict(sample_rate=HZ(390625.0))
GSB_PH = dict(raw=[[S('gsb/sample_gsb_phased.Pol-%s%d.dat' % (p, k)) for k in (1, 2)] for p in 'LR'],
              sample_rate=HZ((1e8 / 3) / 2 ** 23 * 4096 / 512), payload_nbytes=4096)


def shape_probe(name, fmt, path, kw, subsets, fields):
    """Steps: for squeeze on / off and every subset, the sample shape, its named fields and two samples."""
    steps = []
    for squeeze in (True, False):
        for sub in subsets:
            steps += [open_('f', fmt, path, 'rs', squeeze=squeeze, subset=sub, **kw), get('f.subset'),
                      get('f.sample_shape'), get('f.shape')]
            steps += [get('f.sample_shape.' + fl) for fl in fields]
            steps += [call(None, 'f.read', 2), close('f')]
    return steps


CASES = [
    case('threads_and_channels',
         'VDIF sample shapes (nthread, nchan) under squeeze and subset: integers drop an axis, slices and '
         'lists keep it, fields that are gone raise AttributeError '
         '(baseband/base/tests/test_base.py, TestSqueezeAndSubset; vdif/tests/test_vdif.py subset cases)',
         shape_probe('aro', 'vdif', S('sample_arochime.vdif'), ARO,
                     [TUP(), 1, TUP(1, 3), TUP(SL(None), [0, 5, 1023]), TUP([0, 1], SL(0, 1024, 512)), [1]],
                     ('nthread', 'nchan')),
         shape_probe('vdif', 'vdif', S('sample.vdif'), {}, [TUP(), 3, [1, 5], SL(0, 8, 4), TUP(SL(None), 0)],
                     ('nthread', 'nchan'))),

    case('polarisations_and_channels',
         'GUPPI (npol, nchan), DADA (npol,), GSB phased (nthread, nchan): the same rules '
         '(the subset cases of guppi/tests/test_guppi.py, dada/tests/test_dada.py, gsb/tests/test_gsb.py)',
         shape_probe('puppi', 'guppi', S('sample_puppi.raw'), {}, [TUP(), 0, TUP(1, [0, 3]), TUP(SL(None), 2), [0]],
                     ('npol', 'nchan')),
         shape_probe('dada', 'dada', S('sample.dada'), {}, [TUP(), 1, [0, 1], SL(1, 2)], ('npol',)),
         shape_probe('gsb', 'gsb', S('gsb/sample_gsb_phased.timestamp'), GSB_PH,
                     [TUP(), 1, TUP(1, 3), TUP(SL(None), SL(0, 512, 128)), TUP(ARRAY([[1], [0]]), [1, 33, 121, 245])],
                     ('nthread', 'nchan'))),

    case('subsets_that_cannot_index_a_sample',
         'a subset out of range, one that leaves nothing, one with more axes than a sample has: refused at '
         'open with the same message (baseband/base/tests/test_base.py, faulty subsets)',
         [[open_('f', 'vdif', S('sample.vdif'), 'rs', subset=sub, msg_has='cannot be used to properly index')]
          for sub in ([8], TUP(1, 1), TUP(SL(None), 2), SL(9, 12), TUP(0, 0, 0))],
         [[open_('g', 'vdif', S('sample_arochime.vdif'), 'rs', subset=sub, msg_has='cannot be used to properly index', **ARO)]
          for sub in (2, TUP(0, 1024), TUP([0, 1], [0, 1, 2]))],
         open_('h', 'dada', S('sample.dada'), 'rs', subset=TUP(0, 0), msg_has='cannot be used to properly index'),
         open_('i', 'guppi', S('sample_puppi.raw'), 'rs', subset=TUP(0, 4), msg_has='cannot be used to properly index')),
]
