"""Binary readers: header searches and frame reads from seeded random places in the sample recordings and in
damaged copies of them."""
from ._dsl import *    # noqa: F401,F403

FILES = (
    ('vdif', S('sample.vdif'), {}, 16 * 5032, True),
    ('vdif', S('sample_vlbi.vdif'), {}, 16 * 5032, True),
    ('mark5b', S('sample.m5b'), dict(kday=56000, nchan=8, bps=2), 4 * 10016, False),
    ('mark4', S('sample.m4'), dict(ntrack=64, decade=2010), 2 * 160000 + 0xa88, False),
    ('mark4', S('sample_32track.m4'), dict(ntrack=32, decade=2010), 2 * 80000 + 9656, False),
)


def searches(k, fmt, path, kw, size, with_header, n=14):
    steps, x = [open_('fb', fmt, path, 'rb', **kw), do('fb.seek', 0), call('h0', 'fb.find_header'), call(None, 'fb.tell')], 70 + k
    for j in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        pos = (x >> 24) % size
        fwd = bool((x >> 50) & 1)
        steps += [do('fb.seek', pos), call(None, 'fb.find_header', forward=fwd), call(None, 'fb.tell')]
        if j % 3 == 0:
            args = [V('h0')] if with_header else []
            steps += [do('fb.seek', pos), call(None, 'fb.locate_frames', *args, forward=fwd),
                      do('fb.seek', pos), call(None, 'fb.locate_frames', *args, forward=fwd, check=[-1, 1], maximum=3000)]
        if j % 4 == 1:
            steps += [call('fr', 'fb.read_frame'), get('fr.header'), item(None, 'fr', SL(0, 3)), call(None, 'fb.tell')]
    steps += [close('fb')]
    return steps


CASES = [
    case('searches_from_random_places',
         'find_header forward and backward, locate_frames with and without checks either side, and the frame '
         'read at the header found, from fourteen seeded random byte positions in each of five recordings '
         '(the find_header / locate_frames tests of VDIF, Mark 5B and Mark 4, positions widened)',
         [searches(k, *f) for k, f in enumerate(FILES)]),
]
