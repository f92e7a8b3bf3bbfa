"""Round 4: what goes wrong when an arena is trimmed and grown again at once?
A GUPPI channels-first decode (k_decode_i8_xpose) into a block of a freshly
(re)grown step, checked on the device (per-channel sums, as
tests/test_fullsize_gpu.py does) and on the host (a D2H copy of the first 64
MiB against NumPy), for: the same arena trimmed and regrown (same virtual
addresses, new physical memory), a new Arena object per round, and an arena
that keeps its memory."""
import sys, os, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import arena, kernels, _lib
kernels.init()
dev = torch.device('cuda')
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nfr = 8
g = torch.Generator(device=dev); g.manual_seed(17)
image = torch.empty(nfr * blk, device=dev, dtype=torch.uint8)
b = image.view(torch.int8).view(nfr, nchan, T, npol, 2)
want_sum, host_want = [], None


def new_data():
    """other bytes every round: stale output of the round before must not pass"""
    global want_sum, host_want
    image.copy_(torch.randint(0, 256, (nfr * blk,), generator=g, device=dev, dtype=torch.uint8))
    want_sum = [b[f].to(torch.float64).sum(1).permute(1, 0, 2).contiguous() for f in range(nfr)]
    host_want = b[0, :, :4096].permute(1, 2, 0, 3).to(torch.float32).cpu().numpy()      # (4096, npol, nchan, 2)


def check(o):
    o = o.view(nfr, T, npol, nchan, 2)
    bad = [f for f in range(nfr) if not torch.equal(o[f].to(torch.float64).sum(0), want_sum[f])]
    host = o[0, :4096].cpu().numpy()
    return {"frames_wrong_on_device": bad, "host_copy_equal": bool(np.array_equal(host, host_want)),
            "zeros_in_frame0": int((o[0] == 0).sum().item()), "expected_zeros_in_frame0": int((b[0] == 0).sum().item())}


def decode(ar):
    o = ar.empty(nfr * T * npol * nchan * 2)
    kernels.decode_i8_tiled(image, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=o)
    torch.cuda.synchronize()
    return o


mode = sys.argv[1] if len(sys.argv) > 1 else 'same'
ar = arena.Arena(200 << 30)
for rnd in range(4):
    new_data()
    o = decode(ar)
    r = check(o)
    time.sleep(0.5)
    r2 = check(o)
    r.update({"mode": mode, "round": rnd, "ptr": hex(o.data_ptr()), "again_after_0.5s": r2["frames_wrong_on_device"]})
    print(json.dumps(r), flush=True)
    del o
    if mode == 'same':
        ar.trim()
    elif mode == 'new':
        ar.close()
        ar = arena.Arena(200 << 30)
    elif mode == 'same_sleep':
        ar.trim()
        time.sleep(3.0)
