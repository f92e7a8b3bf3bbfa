"""What a read-only pass over float32 reaches on this box, as a yardstick for the encoders
(2-bit: 16 bytes read per byte written): torch reductions, and the 2-bit encoder on the same buffer."""
import sys

import torch

sys.path.insert(0, '.')


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
    n = int(gib * 2 ** 30) // 4
    x = torch.empty(n, dtype=torch.float32, device='cuda').normal_()
    for name, fn in (('torch.sum', lambda: torch.sum(x)), ('torch.amax', lambda: torch.amax(x))):
        ms = timed(fn)
        print('%-22s %.3f ms  %.0f GB/s read = %.3f of 8 TB/s' % (name, ms, n * 4 / ms / 1e6, n * 4 / ms / 1e6 / 8000))
    from baseband_amd import kernels, _lib
    for bps in (2, 4, 8):
        ms = timed(lambda: kernels.encode_flat(x.view(-1, 1), _lib.CODER_VDIF, bps))
        moved = n * 4 + n * bps // 8
        print('encode %d-bit           %.3f ms  %.0f GB/s moved = %.3f of 8 TB/s' % (bps, ms, moved / ms / 1e6, moved / ms / 1e6 / 8000))


if __name__ == '__main__':
    main()
