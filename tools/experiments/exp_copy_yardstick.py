"""What a plain device-to-device copy reaches on this box, as a yardstick for k_copy_frames
(DADA NBIT 32 pass-through: as many bytes read as written).  torch's copy kernel and hipMemcpyAsync."""
import sys
import time

import torch

sys.path.insert(0, '.')


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    n = int(gib * 2 ** 30) // 4
    src = torch.empty(n, dtype=torch.float32, device='cuda').normal_()
    dst = torch.empty_like(src)
    for name, fn in (('torch copy_', lambda: dst.copy_(src)),
                     ('torch clone-like add0', lambda: torch.add(src, 0.0, out=dst))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        print('%-24s %.3f ms  %.0f GB/s moved (read + written) = %.3f of 8 TB/s' % (
            name, ms, 2 * n * 4 / ms / 1e6, 2 * n * 4 / ms / 1e6 / 8000))
    # the product's frame copy on the same amount (DADA-like frames of 64 MiB + 4096-byte headers)
    from baseband_amd import kernels
    frame = (64 << 20) + 4096
    nfr = int(gib * 2 ** 30) // frame
    img = torch.empty(nfr * frame, dtype=torch.uint8, device='cuda')
    out = torch.empty(nfr * (64 << 20) // 4, dtype=torch.float32, device='cuda')
    if hasattr(kernels, 'copy_frames'):
        f = lambda: kernels.copy_frames(img, nfr, 64 << 20, 4096, frame, out=out)
        try:
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(5):
                f()
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 5
            print('%-24s %.3f ms  %.0f GB/s moved = %.3f of 8 TB/s' % ('k_copy_frames', ms, 2 * nfr * (64 << 20) / ms / 1e6,
                                                                     2 * nfr * (64 << 20) / ms / 1e6 / 8000))
        except Exception as exc:
            print('k_copy_frames not run:', repr(exc)[:200])


if __name__ == '__main__':
    main()
