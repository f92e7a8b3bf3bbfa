"""Mark 4 track-demultiplexing maps (data).

For every decoder the reference registers (mark4/payload.py:338-342), keyed
like its ``_decoders`` dict by ``(nchan, bps-or-magnitude-signature, fanout)``:
``sign_bit[t*nchan + c]`` / ``mag_bit[t*nchan + c]`` give the bit of the
``ntrack``-bit stream word that carries the sign / magnitude of fanout sample
``t``, channel ``c``.  The numbers follow from the track assignments of the
Mark 4 memo 230.3 (tables 10-14) and were extracted by pushing single-bit
words through the reference decoders (oracle/gen_golden.py,
tests/golden/mark4_bitmaps.json; tests/test_host_logic.py re-checks them).
"""

FT_SIGNATURE = 0xf0faf050f0faf05

BITMAPS = {
    (16, FT_SIGNATURE, 2): dict(
        ntrack=64,
        sign_bit=[0, 4, 16, 24, 1, 9, 17, 25, 32, 36, 48, 56, 33, 41, 49, 57, 2, 6, 18, 26, 3, 11, 19, 27, 34, 38, 50, 58, 35, 43, 51, 59],
        mag_bit=[8, 12, 20, 28, 5, 13, 21, 29, 40, 44, 52, 60, 37, 45, 53, 61, 10, 14, 22, 30, 7, 15, 23, 31, 42, 46, 54, 62, 39, 47, 55, 63]),
    (2, 2, 4): dict(
        ntrack=16,
        sign_bit=[0, 8, 1, 9, 2, 10, 3, 11],
        mag_bit=[4, 12, 5, 13, 6, 14, 7, 15]),
    (4, 2, 4): dict(
        ntrack=32,
        sign_bit=[0, 16, 1, 17, 2, 18, 3, 19, 4, 20, 5, 21, 6, 22, 7, 23],
        mag_bit=[8, 24, 9, 25, 10, 26, 11, 27, 12, 28, 13, 29, 14, 30, 15, 31]),
    (8, 2, 2): dict(
        ntrack=32,
        sign_bit=[0, 8, 16, 24, 1, 9, 17, 25, 2, 10, 18, 26, 3, 11, 19, 27],
        mag_bit=[4, 12, 20, 28, 5, 13, 21, 29, 6, 14, 22, 30, 7, 15, 23, 31]),
    (8, 2, 4): dict(
        ntrack=64,
        sign_bit=[0, 16, 1, 17, 32, 48, 33, 49, 2, 18, 3, 19, 34, 50, 35, 51, 4, 20, 5, 21, 36, 52, 37, 53, 6, 22, 7, 23, 38, 54, 39, 55],
        mag_bit=[8, 24, 9, 25, 40, 56, 41, 57, 10, 26, 11, 27, 42, 58, 43, 59, 12, 28, 13, 29, 44, 60, 45, 61, 14, 30, 15, 31, 46, 62, 47, 63]),
}
