"""File images that already live in HBM.

north_star: "reads ... frames staged in HBM ... keeping
``baseband.open()/StreamReader.read()`` as the drop-in API".  The reference's
readers take a file name or handle and read a frame at a time
(base/base.py:919-969).  Here every format's ``open()`` also takes a
``torch.uint8`` device tensor holding the file bytes (`DeviceFile` wraps it as
the file handle), and any stream reader opened from a host file can be told to
keep the file in HBM (``fh.stage()``).  ``read()`` on such a reader launches
ONE header scan, ONE index build and ONE decode for the whole request -- no
windows, no host copy; the few bytes the host needs (first header, thread
ids, the last header) come back with small device-to-host copies.
"""
import io

import numpy as np
import torch

__all__ = ['DeviceImage', 'DeviceFile', 'device_bytes_of']


class DeviceImage:
    """The `staging.host_image` protocol over a device tensor: ``len()``,
    slicing (-> NumPy copy of those bytes), ``header_words`` (lazy strided
    header table), plus `device_tensor`, which the readers decode from."""

    def __init__(self, dev):
        if not (isinstance(dev, torch.Tensor) and dev.is_cuda):
            raise TypeError("DeviceImage needs a tensor on the GPU")
        if dev.dtype != torch.uint8:
            dev = dev.contiguous().view(torch.uint8)
        self.device_tensor = dev.reshape(-1)
        if not self.device_tensor.is_contiguous():
            self.device_tensor = self.device_tensor.contiguous()
        self.dtype = np.dtype(np.uint8)

    def __len__(self):
        return self.device_tensor.numel()

    shape = property(lambda self: (len(self),))

    def __getitem__(self, item):
        if not isinstance(item, slice):
            i = int(item)
            if i < 0:
                i += len(self)
            if not 0 <= i < len(self):
                raise IndexError(item)
            return np.uint8(self.device_tensor[i].item())
        lo, hi, step = item.indices(len(self))
        if step != 1:
            raise IndexError("only contiguous slices of a device image")
        if hi <= lo:
            return np.empty(0, np.uint8)
        return self.device_tensor[lo:hi].cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        whole = self[:]
        return whole if dtype in (None, whole.dtype) else whole.astype(dtype)

    def pieces(self, lo, hi):
        """Host copies of bytes [lo, hi) (the staging protocol; a resident
        image is normally decoded in place and never staged)."""
        lo, hi = max(0, lo), min(hi, len(self))
        return [self[lo:hi]] if hi > lo else []

    # -- fixed-stride header table (base.header.strided_header_words)
    def header_words(self, frame_nbytes, nwords, offset=0):
        from .helpers.sequentialfile import _HeaderTable
        return _HeaderTable(self, frame_nbytes, nwords, offset)

    def _gather_headers(self, frame_nbytes, nwords, offset, first_row, last_row):
        """Rows [first_row, last_row) of the header table as a (rows, nwords)
        uint32 array: one strided gather on the device, one small copy."""
        n = max(0, last_row - first_row)
        hb = 4 * nwords
        if n == 0:
            return np.empty((0, nwords), '<u4')
        start = offset + first_row * frame_nbytes
        t = self.device_tensor
        rows = torch.as_strided(t, (n, hb), (frame_nbytes, 1), storage_offset=t.storage_offset() + start)
        return rows.contiguous().cpu().numpy().view('<u4').reshape(n, nwords)


class DeviceFile(io.RawIOBase):
    """Binary file handle over a `DeviceImage`: what ``open(<device tensor>)``
    hands to the file and stream readers.  ``read`` copies the requested bytes
    to the host (headers are a few dozen bytes); `host_image` is the image
    itself, so the readers find the bytes already in HBM."""

    def __init__(self, dev):
        super().__init__()
        self._image = dev if isinstance(dev, DeviceImage) else DeviceImage(dev)
        self._pos = 0
        self.name = '<device image of {} bytes>'.format(len(self._image))

    def host_image(self):
        return self._image

    def readable(self):
        return True

    def seekable(self):
        return True

    def writable(self):
        return False

    def fileno(self):
        raise OSError("a device image has no file descriptor")

    def tell(self):
        return self._pos

    def seek(self, offset, whence=0):
        if whence == 0:
            pos = offset
        elif whence == 1:
            pos = self._pos + offset
        elif whence == 2:
            pos = len(self._image) + offset
        else:
            raise ValueError("invalid whence ({})".format(whence))
        if pos < 0:
            raise OSError("negative seek position {}".format(pos))
        self._pos = pos
        return pos

    def read(self, size=-1):
        n = len(self._image)
        lo = min(self._pos, n)
        hi = n if size is None or size < 0 else min(n, lo + size)
        self._pos = hi
        return self._image[lo:hi].tobytes()

    def readinto(self, b):
        data = self.read(len(b))
        b[:len(data)] = data
        return len(data)

    def readline(self, size=-1):
        """Up to and including the next newline (GSB timestamp files are host
        files; this exists for completeness of the handle)."""
        n = len(self._image)
        out = bytearray()
        while self._pos < n and (size < 0 or len(out) < size):
            chunk = self._image[self._pos:min(n, self._pos + 256)].tobytes()
            k = chunk.find(b'\n')
            if k >= 0:
                chunk = chunk[:k + 1]
            if size >= 0:
                chunk = chunk[:size - len(out)]
            out += chunk
            self._pos += len(chunk)
            if out.endswith(b'\n'):
                break
        return bytes(out)


def device_bytes_of(image):
    """The device tensor behind `image` when it is resident in HBM, else None."""
    return getattr(image, 'device_tensor', None)
