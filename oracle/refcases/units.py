"""Positions of streams in other units than samples: tell(unit), seek by durations and instants."""
from ._dsl import *    # noqa: F401,F403

OPENERS = (
    ('vdif', S('sample.vdif'), {}),
    ('mark5b', S('sample.m5b'), dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2)),
    ('mark4', S('sample.m4'), dict(ntrack=64, decade=2010)),
    ('dada', S('sample.dada'), {}),
    ('guppi', S('sample_puppi.raw'), {}),
)

CASES = [
    case('tell_in_units_and_seek_by_time',
         'every stream reader: tell() in seconds, milliseconds and as an instant at several offsets; seeks '
         'by a duration from the start, the current position and the end, and by an instant; positions '
         'that are not a whole number of samples are rounded as the reference rounds them '
         '(base/tests and every format: tell(unit=...) / seek(Quantity | Time) cases)',
         [[open_('fh', fmt, path, 'rs', **kw), get('fh.sample_rate'), get('fh.start_time', as_='t0'),
           [[do('fh.seek', n), call(None, 'fh.tell', unit=UNIT('s')), call(None, 'fh.tell', unit=UNIT('ms')),
             call(None, 'fh.tell', unit='time')] for n in (0, 1, 1001, 4097)],
           call(None, 'fh.seek', NS(31250)), call(None, 'fh.tell'),
           call(None, 'fh.seek', NS(31250), 1), call(None, 'fh.tell'),
           call(None, 'fh.seek', NS(-62500), 'end'), call(None, 'fh.tell'),
           call(None, 'fh.seek', NS(15), 0), call(None, 'fh.seek', NS(16), 0), call(None, 'fh.seek', NS(47), 0),
           fn('later', 'add', V('t0'), NS(93750), quiet=True), call(None, 'fh.seek', V('later')), get('fh.time'),
           call(None, 'fh.tell', unit='bla'),
           call(None, 'fh.seek', NS(-1), 0), call(None, 'fh.read', 1),
           close('fh')]
          for fmt, path, kw in OPENERS]),

    case('writers_tell_time',
         'stream writers count samples and report the time of the next one (every format: fw.tell(unit), '
         'fw.time while writing)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read', 20000),
         open_('fw', 'vdif', T('w.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         call(None, 'fw.tell'), call(None, 'fw.tell', unit='time'), call(None, 'fw.tell', unit=UNIT('us')),
         item('p', 'd', SL(0, 12345), quiet=True), do('fw.write', V('p')),
         call(None, 'fw.tell'), call(None, 'fw.tell', unit='time'), call(None, 'fw.tell', unit=UNIT('us')), get('fw.time'),
         item('q', 'd', SL(12345, 20000), quiet=True), do('fw.write', V('q')), call(None, 'fw.tell', unit='time'),
         close('fw'), close('fr'), digest(T('w.vdif'))),
]
