"""Stream writers fed in uneven pieces, with stretches flagged invalid: the bytes on disk."""
from ._dsl import *    # noqa: F401,F403

L2 = [-3.316505, -1.0, 1.0, 3.316505]
L4 = [-7.0, -2.5, -1.0, -0.3, 0.0, 0.4, 1.0, 2.2, 6.9]
L8 = [-120.0, -35.0, -3.6, -0.5, 0.0, 0.7, 3.3, 50.0, 126.0]


def pieces(var, cuts, flags):
    """Steps: write `var` in slices [cuts[k], cuts[k+1]) with the validity flags given."""
    steps = []
    for k in range(len(cuts) - 1):
        steps += [item('part', var, SL(cuts[k], cuts[k + 1]), quiet=True), do('fw.write', V('part'), valid=flags[k])]
    return steps


CASES = [
    case('vdif_threads_and_widths',
         'VDIF writers for 2-bit complex 8 threads x 16 channels (the cfg3 layout), 4-bit real, 8-bit '
         'complex and 1-bit: data written in uneven pieces, some flagged invalid; file digests, then what '
         'a reader returns (vdif/tests/test_vdif.py, test_stream_writer and the bps / complex variants)',
         [[let('d', RNG(seed, shape, levels, complex=cplx)),
           open_('fw', 'vdif', T(name), 'ws', sample_rate=HZ(rate), nthread=shape[1], nchan=shape[2], bps=bps,
                 complex_data=cplx, edv=edv, station='ab', samples_per_frame=spf, time=TIME('2020-02-29T23:59:59'),
                 squeeze=False),
           get('fw.header0'), get('fw.sample_shape'),
           pieces('d', cuts, flags), call(None, 'fw.tell'), get('fw.time'), close('fw'), digest(T(name)),
           open_('fr', 'vdif', T(name), 'rs', squeeze=False, **({} if edv in (1, 3) else dict(sample_rate=HZ(rate)))),
           get('fr.shape'), get('fr.start_time'), get('fr.stop_time'), call(None, 'fr.read'), close('fr')]
          for seed, name, shape, levels, cplx, bps, edv, spf, rate, cuts, flags in (
              (1, 'c3.vdif', (4000, 8, 16), L2, True, 2, 0, 1000, 1e6, (0, 1, 999, 1000, 2500, 3001, 4000),
               (True, True, True, False, True, True)),
              (2, 'b4.vdif', (960, 2, 4), L4, False, 4, 1, 240, 48000., (0, 100, 480, 481, 960), (True, False, True, True)),
              (3, 'b8.vdif', (512, 1, 2), L8, True, 8, 1, 128, 128000., (0, 128, 300, 512), (False, True, True)),
              (4, 'b1.vdif', (2048, 1, 8), [-1.0, 1.0], False, 1, 0, 512, 512000., (0, 700, 2048), (True, True)))]),

    case('mark5b_and_mark4_invalid_stretches',
         'Mark 5B marks invalid frames by a fill pattern in the payload, Mark 4 by error bits in the '
         'header: writers fed pieces with invalid stretches, digests, and what readers make of the files '
         '(mark5b/tests/test_mark5b.py and mark4/tests/test_mark4.py, stream writer tests with valid=False)',
         let('d5', RNG(7, (4 * 5000, 8), L2)),
         open_('fw', 'mark5b', T('v.m5b'), 'ws', sample_rate=HZ(32e6), nchan=8, bps=2, time=TIME('2014-06-13T05:30:01')),
         pieces('d5', (0, 4999, 5000, 5001, 12000, 20000), (True, True, False, True, True)), close('fw'),
         digest(T('v.m5b')),
         open_('fr', 'mark5b', T('v.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2, fill_value=-9.0),
         get('fr.shape'), call('back', 'fr.read'), item(None, 'back', SL(4995, 5005)), item(None, 'back', SL(9995, 10005)),
         close('fr'),
         open_('fb', 'mark5b', T('v.m5b'), 'rb', kday=56000, nchan=8, bps=2), do('fb.seek', 10016),
         call('f1', 'fb.read_frame'), get('f1.valid'), item(None, 'f1.payload.words', SL(0, 4)), close('fb'),
         let('d4', RNG(8, (3 * 40000, 8), L2)),
         open_('fw', 'mark4', T('v.m4'), 'ws', sample_rate=HZ(16e6), ntrack=32, fanout=2, nchan=8, bps=2,
               time=TIME('2014-06-16T07:38:12.4750')),
         get('fw.header0'), get('fw.samples_per_frame'),
         pieces('d4', (0, 39999, 40001, 90000, 120000), (True, False, True, True)), close('fw'), digest(T('v.m4')),
         open_('fr', 'mark4', T('v.m4'), 'rs', sample_rate=HZ(16e6), ntrack=32, decade=2010, fill_value=-9.0),
         get('fr.shape'), get('fr.start_time'), call('back', 'fr.read'), item(None, 'back', SL(318, 324)),
         item(None, 'back', SL(39998, 40004)), item(None, 'back', SL(80318, 80324)), close('fr')),

    case('block_formats_in_pieces',
         'DADA and GUPPI writers fed sample by sample counts that do not divide the block: digests '
         '(dada/tests/test_dada.py and guppi/tests/test_guppi.py, stream writer tests)',
         let('dd', RNG(9, (3000, 2, 4), L8, complex=True)),
         open_('fw', 'dada', T('p.dada'), 'ws', time=TIME('2013-07-02T01:39:20'), sample_rate=HZ(16e6),
               samples_per_frame=1000, npol=2, nchan=4, bps=8, complex_data=True),
         get('fw.header0'), pieces('dd', (0, 7, 1000, 1999, 3000), (True, True, True, True)), close('fw'),
         digest(T('p.dada')),
         open_('fr', 'dada', T('p.dada'), 'rs'), get('fr.shape'), call('b', 'fr.read'), eq(V('b'), V('dd')), close('fr'),
         let('dg', RNG(10, (2048, 2, 8), L8, complex=True)),
         open_('fg', 'guppi', S('sample_puppi.raw'), 'rs'), call('hg', 'fg.header0.copy'), close('fg'),
         set_('hg.overlap', 0), set_('hg.nchan', 8) , set_('hg.samples_per_frame', 512),
         open_('fw', 'guppi', T('p.raw'), 'ws', header0=V('hg')),
         pieces('dg', (0, 500, 513, 2048), (True, True, True)), close('fw'), digest(T('p.raw')),
         open_('fr', 'guppi', T('p.raw'), 'rs'), get('fr.shape'), call('b', 'fr.read'), eq(V('b'), V('dg')), close('fr')),
]
