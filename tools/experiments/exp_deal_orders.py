"""The SAME physical memory -- 48 GiB created as 1536 chunks of 32 MiB -- mapped in different
orders at fresh addresses, one after the other, and cfg2 decodes of 2^15 frames (4.2 GB) and
8000 frames (1 GB) timed into blocks of each mapping.  Separates WHERE the memory lies
(fixed here) from HOW a block's granules are dealt over it:
  creation   granule k = chunk k (what one plain allocation gives: physically contiguous runs)
  dealt      the product's order: consecutive granules from consecutive 1 GiB teeth
  skewed     dealt, and the position inside the tooth shifted by 7 per tooth
  random     a fixed pseudo-random permutation of all chunks
  dealt8     dealt over 8 GiB super-teeth (6 of them)
    python tools/experiments/exp_deal_orders.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib                  # noqa: E402

hip = C.CDLL('libamdhip64.so')
GIB, CHUNK, TEETH = 1 << 30, 32 << 20, 48
CPT = GIB // CHUNK
N = TEETH * CPT


class Prop(C.Structure):
    _fields_ = [('type', C.c_int), ('requestedHandleType', C.c_int), ('location_type', C.c_int), ('location_id', C.c_int),
                ('win32', C.c_void_p), ('allocFlags', C.c_ubyte * 8)]


class Access(C.Structure):
    _fields_ = [('location_type', C.c_int), ('location_id', C.c_int), ('flags', C.c_int)]


def ok(rc, what):
    if rc != 0:
        raise RuntimeError('%s: hip error %d' % (what, rc))


dev = torch.device('cuda', 0)
kernels.init()
nframes = (2 << 30) // bench.FRAME_NBYTES
image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
prop = Prop()
prop.type, prop.location_type, prop.location_id = 1, 1, 0      # pinned, device 0
handles = []
for k in range(N):
    h = C.c_void_p()
    ok(hip.hipMemCreate(C.byref(h), C.c_size_t(CHUNK), C.byref(prop), C.c_ulonglong(0)), 'hipMemCreate')
    handles.append(h)
base = C.c_void_p()
ok(hip.hipMemAddressReserve(C.byref(base), C.c_size_t(8 * N * CHUNK), C.c_size_t(GIB), None, C.c_ulonglong(0)), 'reserve')
acc = Access()
acc.location_type, acc.location_id, acc.flags = 1, 0, 3


class Raw:
    def __init__(self, ptr, nfloat):
        self.__cuda_array_interface__ = {'shape': (nfloat,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2, 'strides': None}


def rate(out, nf, first):
    ts = []
    for r in range(5):
        win = image[((first + r * 7919) % (nframes - nf)) * bench.FRAME_NBYTES:][:nf * bench.FRAME_NBYTES]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32, src_stride=bench.FRAME_NBYTES, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / float(np.median(ts)) / 1e6


rng = np.random.default_rng(5)
orders = {
    'creation': list(range(N)),
    'dealt': [(k % TEETH) * CPT + k // TEETH for k in range(N)],
    'skewed': [(k % TEETH) * CPT + (k // TEETH + 7 * (k % TEETH)) % CPT for k in range(N)],
    'random': [int(x) for x in rng.permutation(N)],
    'dealt8': [(k % 6) * (8 * CPT) + k // 6 for k in range(N)],
}
slot = 0
for rep in range(2):
    for name, order in orders.items():
        assert sorted(order) == list(range(N))
        va = base.value + slot * N * CHUNK          # fresh addresses every time (never reuse a mapping's)
        slot += 1
        for g, c in enumerate(order):
            ok(hip.hipMemMap(C.c_void_p(va + g * CHUNK), C.c_size_t(CHUNK), C.c_size_t(0), handles[c], C.c_ulonglong(0)), 'map')
        ok(hip.hipMemSetAccess(C.c_void_p(va), C.c_size_t(N * CHUNK), C.byref(acc), C.c_size_t(1)), 'access')
        r4, r1 = [], []
        for k in range(8):
            t = torch.as_tensor(Raw(va + k * 5 * GIB, (1 << 15) * bench.SPF), device=dev)
            r4.append(rate(t, 1 << 15, 31 * k))
        for k in range(8):
            t = torch.as_tensor(Raw(va + (k * 5 + 4) * GIB, 8000 * bench.SPF), device=dev)
            r1.append(rate(t, 8000, 17 * k))
        torch.cuda.synchronize()
        for g in range(N):
            ok(hip.hipMemUnmap(C.c_void_p(va + g * CHUNK), C.c_size_t(CHUNK)), 'unmap')
        print("%-9s 4.2 GB blocks: median %.0f (%s) | 1 GB blocks: median %.0f (%s)" % (
            name, np.median(r4), " ".join("%.0f" % x for x in r4), np.median(r1), " ".join("%.0f" % x for x in r1)), flush=True)
