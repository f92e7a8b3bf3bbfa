"""Byte slips of a damaged recording, by frame number.

The reference's stream readers keep a `RawOffsets` table of how far the frames of a
file lie from where a fixed stride would put them, filled in as `_bad_frame` meets the
damage (base/offsets.py:6-126; base/base.py:1064,1081,1189; vdif/base.py:615,729).
Here the whole file is located at once on the GPU (`_relocate`), which leaves a dense
index frame -> byte position in HBM; `RawOffsets.from_index` folds such an index into
the same minimal step table, and readers show it as `fh._raw_offsets`.

The class answers as the reference's does: ``table[frame_nr]`` is the position of a
frame (slip + frame_nr * frame_nbytes), ``table[frame_nr] = position`` records one,
and the lists `frame_nr` / `offset` stay as short as the information allows.
"""
import operator
from bisect import bisect_right

import numpy as np

__all__ = ['RawOffsets']


class RawOffsets:
    """Step function frame number -> slip in bytes (0 before the first step).

    Parameters
    ----------
    frame_nr, offset : list, optional
        First frame number each slip holds from, and the slips.
    frame_nbytes : int
        Stride added to what is returned: ``slip + frame_nr * frame_nbytes``.
    """

    def __init__(self, frame_nr=None, offset=None, frame_nbytes=0):
        frame_nr = [] if frame_nr is None else frame_nr
        offset = [] if offset is None else offset
        if len(frame_nr) != len(offset):
            raise ValueError('must have equal number of frame numbers and offsets.')
        self.frame_nr = frame_nr
        self.offset = offset
        self.frame_nbytes = operator.index(frame_nbytes)

    @classmethod
    def from_index(cls, position, frame_nbytes, first=0, known=None):
        """The table of a dense index: `position[i]` is where frame `i` starts in
        the file; frames that are missing (`known` false; by default a negative
        position) tell nothing and take the slip of the frame before them, as a
        frame the reference never visited would."""
        position = np.asarray(position, dtype=np.int64)
        frame_nbytes = operator.index(frame_nbytes)
        known = np.nonzero(position >= 0 if known is None else np.asarray(known, bool))[0]
        slip = position[known] - (known + first) * frame_nbytes
        if slip.size == 0:
            return cls(frame_nbytes=frame_nbytes)
        step = np.nonzero(np.diff(slip, prepend=0))[0]
        return cls([int(f) for f in known[step] + first], [int(s) for s in slip[step]],
                   frame_nbytes)

    def _step(self, frame_nr):
        """Number of steps at or before `frame_nr`."""
        return bisect_right(self.frame_nr, frame_nr)

    def _slip(self, k):
        return self.offset[k - 1] if k > 0 else 0

    def __getitem__(self, frame_nr):
        expected = frame_nr * self.frame_nbytes
        return expected + self._slip(self._step(frame_nr)) if self.frame_nr else expected

    def __setitem__(self, frame_nr, position):
        slip = position - frame_nr * self.frame_nbytes
        k = self._step(frame_nr)
        if k and self.frame_nr[k - 1] == frame_nr:
            # a step sits exactly here: drop it and decide again, the new value may
            # agree with a neighbour
            if self.offset[k - 1] == slip:
                return
            k -= 1
            del self.frame_nr[k], self.offset[k]
        if k < len(self.frame_nr) and self.offset[k] == slip:
            self.frame_nr[k] = frame_nr         # the next step starts earlier than thought
        elif slip != self._slip(k):
            self.frame_nr.insert(k, frame_nr)
            self.offset.insert(k, slip)

    def __len__(self):
        return len(self.frame_nr)

    def __repr__(self):
        return '{}(frame_nr={}, offset={}, frame_nbytes={})'.format(
            type(self).__name__, self.frame_nr, self.offset, self.frame_nbytes)
