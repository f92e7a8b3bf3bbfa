"""Item access on payloads of every format: what index forms are taken, what an assignment re-encodes,
what is refused."""
from ._dsl import *    # noqa: F401,F403

LEVELS2 = [-3.316505, -1.0, 1.0, 3.316505]
WIDE = [-9.0, -3.316505, -2.0, -1.0, -0.3, 0.0, 0.4, 1.0, 2.2, 3.316505, 50.0]


def walk(p, nsample, trailing):
    """The same tour of a payload `p` with `nsample` samples whose samples have `trailing` (a tuple)."""
    inner = tuple(0 for _ in trailing)
    steps = [get(p + '.shape'), get(p + '.dtype'), get(p + '.nbytes'), get(p + '.size'), get(p + '.ndim'),
             item(None, p, 0), item(None, p, -1), item(None, p, nsample - 1), item(None, p, nsample),
             item(None, p, -nsample - 1),
             item(None, p, SL(None)), item(None, p, SL(3, 11)), item(None, p, SL(-7, None)), item(None, p, SL(5, 5)),
             item(None, p, SL(2, 20, 3)), item(None, p, SL(None, None, -1)),
             item(None, p, TUP(SL(4, 9)) ), item(None, p, TUP(7) + () if False else TUP(7)),
             item(None, p, TUP(SL(4, 9), *inner)), item(None, p, TUP(6, *inner)),
             item(None, p, TUP(ELLIPSIS, *inner[-1:])), item(None, p, [1, 5, 6]),
             item(None, p, TUP(SL(0, 4), *(len(trailing) * [SL(None)]), 0))]
    if trailing and trailing[-1] > 1:
        steps += [item(None, p, TUP(SL(8, 12), *inner[:-1], SL(0, trailing[-1], 2))),
                  item(None, p, TUP(SL(8, 12), *inner[:-1], -1)),
                  item(None, p, TUP(3, *inner[:-1], trailing[-1]))]
    return steps


def rewrite(p, nsample, trailing, cplx=False):
    one = -1.0
    row = [[LEVELS2[(i + j) % 4] for j in range(trailing[-1])] for i in range(4)] if len(trailing) == 1 else None
    steps = [setitem(p, 2, one), item(None, p, SL(0, 5)), get(p),
             setitem(p, SL(4, 8), 1.0), item(None, p, SL(3, 9)),
             setitem(p, SL(None), RNG(77, (nsample,) + tuple(trailing), WIDE, complex=cplx)), get(p), get(p + '.data'),
             setitem(p, SL(10, 14), RNG(78, (4,) + tuple(trailing), LEVELS2, complex=cplx)), item(None, p, SL(9, 15)), get(p),
             setitem(p, SL(10, 14), RNG(79, (5,) + tuple(trailing), LEVELS2, complex=cplx)),
             setitem(p, nsample, 1.0), setitem(p, SL(2, 12, 2), 1.0), get(p)]
    if row is not None:
        steps += [setitem(p, SL(20, 24), ARRAY(row, 'f4')), item(None, p, SL(19, 25)),
                  setitem(p, TUP(SL(20, 24), 0), ARRAY([1.0, -1.0, 1.0, -1.0], 'f4')), item(None, p, SL(19, 25)),
                  setitem(p, TUP(SL(30, 34), SL(0, 1)), 3.316505), item(None, p, SL(29, 35)), get(p),
                  setitem(p, TUP(40, 0), -1.0), setitem(p, TUP(41, 0), ARRAY([[1.0]], 'f4')),
                  setitem(p, TUP(42, 0), ARRAY([1.0, -1.0], 'f4')), setitem(p, 43, ARRAY([[[1.0]]], 'f4')),
                  item(None, p, SL(39, 45)), get(p)]
    return steps


CASES = [
    case('vdif_payload_items',
         'indexing and assignment on VDIF payloads of 1, 2, 4, 8 and 16 bits, real and complex (vdif/tests/'
         'test_vdif.py, test_payload_getitem_setitem; base/tests/test_base.py payload item cases)',
         [[let('d', RNG(100 + bps, (nsample, nchan), LEVELS2 if bps <= 2 else WIDE, complex=cplx)),
           call('h', 'vdif.VDIFHeader.fromvalues', edv=0, bps=bps, nchan=nchan, complex_data=cplx,
                samples_per_frame=nsample, station='aa', time=TIME('2015-01-01T00:00:00'), frame_rate=HZ(100.)),
           call('p', 'vdif.VDIFPayload.fromdata', V('d'), V('h')), get('p')]
          + walk('p', nsample, (nchan,)) + rewrite('p', nsample, (nchan,), cplx)
          for bps, nchan, cplx, nsample in ((2, 4, False, 64), (1, 8, False, 64), (4, 2, True, 64), (8, 1, False, 64),
                                            (4, 1, False, 64), (2, 1, True, 64))]),

    case('mark5b_payload_items',
         'the same on the 10000-byte Mark 5B payload, 2-bit 8 channels and 1-bit 4 channels '
         '(mark5b/tests/test_mark5b.py, test_payload_getitem_setitem)',
         [[let('d', RNG(200 + bps, (80000 // bps // nchan, nchan), LEVELS2)),
           call('p', 'mark5b.Mark5BPayload.fromdata', V('d'), bps=bps)]
          + walk('p', 80000 // bps // nchan, (nchan,))[:-3] + rewrite('p', 80000 // bps // nchan, (nchan,))
          for bps, nchan in ((2, 8), (1, 4))]),

    case('mark4_payload_items',
         'Mark 4 payloads address samples behind the header gap: 64 tracks fan-out 4 and 32 tracks fan-out 2 '
         '(mark4/tests/test_mark4.py, test_payload_getitem_setitem)',
         [[call('h', 'mark4.Mark4Header.fromvalues', ntrack=ntrack, fanout=fanout, bps=2, decade=2010,
                time=TIME('2014-06-16T07:38:12.47500'), nchan=nchan),
           let('d', RNG(300 + ntrack, (nsample, nchan), LEVELS2)),
           call('p', 'mark4.Mark4Payload.fromdata', V('d'), V('h')), get('p.shape'), get('p.nbytes'),
           item(None, 'p', 0), item(None, 'p', -1), item(None, 'p', SL(3, 11)), item(None, 'p', SL(-7, None)),
           item(None, 'p', TUP(SL(4, 9), 0)), item(None, 'p', TUP(SL(8, 12), SL(0, nchan, 2))), item(None, 'p', nsample),
           item(None, 'p', SL(2, 20, 3)),
           setitem('p', 2, -1.0), item(None, 'p', SL(0, 5)),
           setitem('p', SL(100, 104), RNG(78, (4, nchan), LEVELS2)), item(None, 'p', SL(99, 105)),
           setitem('p', TUP(SL(200, 204), 3), ARRAY([1.0, -1.0, 3.316505, -3.316505], 'f4')), item(None, 'p', SL(199, 205)),
           setitem('p', SL(None), RNG(77, (nsample, nchan), WIDE)), get('p'), get('p.data'),
           setitem('p', SL(10, 14), RNG(79, (5, nchan), LEVELS2)), setitem('p', nsample, 1.0)]
          for ntrack, fanout, nchan, nsample in ((64, 4, 8, 80000 - 640), (32, 2, 8, 40000 - 320), (32, 4, 4, 80000 - 640)
                                                 )]),

    case('dada_and_guppi_payload_items',
         'byte formats: 8-bit complex two polarisations (DADA), 8-bit complex with channels, channel-first '
         'and time-first (GUPPI) (dada/tests/test_dada.py, guppi/tests/test_guppi.py payload item cases)',
         call('hd', 'dada.DADAHeader.fromvalues', time=TIME('2013-07-02T01:37:40'), npol=2, bps=8,
              payload_nbytes=4 * 96, sample_rate=HZ(16e6), nchan=1, complex_data=True),
         let('dd', RNG(401, (96, 2, 1), [-128.0, -7.0, 0.0, 5.0, 127.0], complex=True)),
         call('pd', 'dada.DADAPayload.fromdata', V('dd'), V('hd')), get('pd'),
         walk('pd', 96, (2, 1)),
         setitem('pd', SL(4, 8), ARRAY([[[1 + 2j], [3 - 4j]]] * 4, 'c8')), item(None, 'pd', SL(3, 9)),
         setitem('pd', TUP(SL(10, 12), 1, 0), ARRAY([200.0 + 0j, -200.0 - 3.6j], 'c8')), item(None, 'pd', SL(9, 13)), get('pd'),
         setitem('pd', SL(4, 8), ARRAY([[[1 + 2j], [3 - 4j]]] * 3, 'c8')),
         [[call('hg', 'guppi.GUPPIHeader.fromvalues', time=TIME('2018-01-14T14:11:33'), sample_rate=HZ(250e3),
                samples_per_frame=64, overlap=0, npol=2, nchan=4, bps=8, pktfmt=fmt),
           let('dg', RNG(402, (64, 2, 4), [-128.0, -7.0, 0.0, 5.0, 127.0], complex=True)),
           call('pg', 'guppi.GUPPIPayload.fromdata', V('dg'), V('hg')), get('pg')]
          + walk('pg', 64, (2, 4))
          + [setitem('pg', SL(4, 8), 1.0), item(None, 'pg', SL(3, 9)),
             setitem('pg', TUP(SL(10, 12), 1, SL(1, 3)), ARRAY([[1 + 1j, 2 + 2j], [3 + 3j, 4 + 4j]], 'c8')),
             item(None, 'pg', SL(9, 13)), get('pg'),
             setitem('pg', SL(0, 2), RNG(403, (3, 2, 4), [1.0, 2.0], complex=True))]
          for fmt in ('1SFA', 'SIMPLE')]),

    case('gsb_payload_items',
         'GSB: 4-bit real (rawdump) and 8-bit complex with channels (phased) (gsb/tests/test_gsb.py payload cases)',
         let('d4', RNG(501, (512, 1), [-8.0, -3.0, -1.0, 0.0, 2.0, 7.0])),
         call('p4', 'gsb.GSBPayload.fromdata', V('d4'), bps=4), walk('p4', 512, (1,)),
         setitem('p4', SL(4, 8), 7.0), setitem('p4', 9, -20.0), setitem('p4', 10, 20.0), item(None, 'p4', SL(3, 12)), get('p4'),
         let('d8', RNG(502, (16, 2, 8), [-128.0, -5.0, 0.0, 3.0, 127.0], complex=True)),
         call('p8', 'gsb.GSBPayload.fromdata', V('d8'), bps=8), walk('p8', 16, (2, 8)),
         setitem('p8', TUP(3, 1, SL(2, 5)), ARRAY([1 + 1j, 2 - 2j, -3 + 3j], 'c8')), item(None, 'p8', 3), get('p8')),
]
