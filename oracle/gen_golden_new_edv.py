#!/opt/conda/bin/python3.9
"""Reference answers for user-defined VDIF EDVs (the reference's metaclass registry,
vdif/header.py:39-79, as docs/tutorials/new_edv.rst uses it) and for the sample files
the other generators do not touch (sample_vlbi.vdif, sample_drao_corrupted.vdif,
sample_vegas.raw, sample_blc.raw).  Run in the development container:

    /opt/conda/bin/python3.9 oracle/gen_golden_new_edv.py

writes tests/golden/new_edv_cases.json (data only: words, values, digests)."""
import hashlib
import json
import os
import sys

import numpy as np
np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

import astropy.units as u                                   # noqa: E402
from baseband import vdif, guppi                            # noqa: E402
from baseband import base as vlbi                           # noqa: E402
from baseband.data import (SAMPLE_DRAO_CORRUPT, SAMPLE_VLBI_VDIF, SAMPLE_VDIF, SAMPLE_VEGAS, SAMPLE_BLC)   # noqa: E402

out = {"made_by": "oracle/gen_golden_new_edv.py"}


class VDIFHeader4(vdif.header.VDIFHeader):
    _edv = 4
    _header_parser = vlbi.header.HeaderParser(
        (('invalid_data', (0, 31, 1, False)), ('legacy_mode', (0, 30, 1, False)), ('seconds', (0, 0, 30)),
         ('_1_30_2', (1, 30, 2, 0x0)), ('ref_epoch', (1, 24, 6)), ('frame_nr', (1, 0, 24, 0x0)),
         ('vdif_version', (2, 29, 3, 0x1)), ('lg2_nchan', (2, 24, 5)), ('frame_length', (2, 0, 24)),
         ('complex_data', (3, 31, 1)), ('bits_per_sample', (3, 26, 5)), ('thread_id', (3, 16, 10, 0x0)),
         ('station_id', (3, 0, 16)), ('edv', (4, 24, 8)), ('validity_mask_length', (4, 16, 8, 0)),
         ('sync_pattern', (5, 0, 32, 0xACABFEED)), ('validity_mask', (6, 0, 64, 0))))


h = vdif.header.VDIFHeader.fromvalues(edv=4, seconds=14363767, nchan=1, samples_per_frame=1024, station=65532,
                                      bps=2, complex_data=False, thread_id=3, validity_mask_length=60,
                                      validity_mask=(1 << 59) + 1)
out["edv4"] = {"class": type(h).__name__, "words": [int(w) for w in h.words], "keys": list(h.keys()),
               "values": {k: int(h[k]) for k in h.keys()}, "nbytes": h.nbytes,
               "samples_per_frame": h.samples_per_frame, "station": h.station}


class VDIFHeader4Enhanced(vdif.header.VDIFBaseHeader):
    _edv = 42
    _header_parser = (vdif.header.VDIFBaseHeader._header_parser
                      | vlbi.header.HeaderParser((('validity_mask_length', (4, 16, 8, 0)),
                                                  ('sync_pattern', (5, 0, 32, 0xACABFEED)),
                                                  ('validity_mask', (6, 0, 64, 0)))))
    _properties = vdif.header.VDIFBaseHeader._properties + ('validity',)

    def verify(self):
        super().verify()
        assert 1 <= self['validity_mask_length'] <= 64

    @property
    def validity(self):
        bitmask = np.unpackbits(self['validity_mask'].astype('>u8').view('u1'))[::-1].astype(bool)
        return bitmask[:self['validity_mask_length']]

    @validity.setter
    def validity(self, validity):
        bitmask = np.zeros(64, dtype=bool)
        bitmask[:len(validity)] = validity
        self['validity_mask_length'] = len(validity)
        self['validity_mask'] = np.packbits(bitmask[::-1]).view('>u8')


validity = [True] * 8 + [False] * 45 + [True] * 7
h = vdif.header.VDIFHeader.fromvalues(edv=42, seconds=14363767, nchan=1, samples_per_frame=1024, station=65532,
                                      bps=2, complex_data=False, thread_id=3, validity=validity)
out["edv42"] = {"class": type(h).__name__, "words": [int(w) for w in h.words], "validity_in": validity,
                "validity_out": [bool(x) for x in h.validity], "validity_mask": int(h['validity_mask']),
                "validity_mask_length": int(h['validity_mask_length'])}
try:
    class Again(vdif.header.VDIFBaseHeader):
        _edv = 42
    out["duplicate"] = "accepted"
except ValueError as exc:
    out["duplicate"] = str(exc)
try:
    class NoEDV(vdif.header.VDIFBaseHeader):
        pass
    out["no_edv"] = "accepted"
except ValueError as exc:
    out["no_edv"] = str(exc)

# ---- replacing a class: the DRAO file
vdif.header.VDIF_HEADER_CLASSES.pop(0)


class DRAOVDIFHeaderEnhanced(vdif.header.VDIFHeader0):
    _header_parser = (vdif.header.VDIFHeader0._header_parser
                      | vlbi.header.HeaderParser((('link', (3, 16, 4)), ('slot', (3, 20, 6)), ('eud2', (5, 0, 32)))))

    def __init__(self, words, edv=None, verify=True, **kwargs):
        super().__init__(words, verify=False, **kwargs)
        self.mutable = True
        self['bits_per_sample'] = 3

    def verify(self):
        pass


with vdif.open(SAMPLE_DRAO_CORRUPT, 'rb') as fh:
    frames = []
    for _ in range(3):
        fr = fh.read_frame()
        frames.append({"class": type(fr.header).__name__, "eud2": int(fr.header['eud2']), "link": int(fr.header['link']),
                       "slot": int(fr.header['slot']), "bps": fr.header.bps, "nchan": fr.header.nchan,
                       "complex_data": bool(fr.header.complex_data), "frame_nbytes": fr.header.frame_nbytes,
                       "samples_per_frame": fr.header.samples_per_frame, "shape": list(fr.data.shape),
                       "dtype": str(fr.data.dtype), "sha256": hashlib.sha256(np.ascontiguousarray(fr.data).tobytes()).hexdigest(),
                       "first": np.ascontiguousarray(fr.data[:2]).view(np.float32).reshape(-1)[:8].tolist(),
                       "tell": fh.tell()})
out["drao"] = frames
vdif.header.VDIF_HEADER_CLASSES.pop(0)
vdif.header.VDIF_HEADER_CLASSES[0] = vdif.header.VDIFHeader0

# ---- sample_vlbi.vdif: sample.vdif with uncorrected time stamps (vdif/tests/test_vdif.py:1319-1334)
with vdif.open(SAMPLE_VLBI_VDIF, 'rs') as fh, vdif.open(SAMPLE_VDIF, 'rs') as fc:
    a, b = fh.read(), fc.read()
    out["vlbi"] = {"sample_rate_Hz": float(fh.sample_rate.to_value(u.Hz)), "shape": list(fh.shape),
                   "start_time": fh.start_time.isot, "stop_time_ns_after_start": float(((fh.stop_time - fh.start_time).to_value(u.ns))),
                   "same_start_as_sample_vdif": bool(fh.start_time == fc.start_time),
                   "same_samples_as_sample_vdif": bool(np.all(a == b)),
                   "sha256": hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(),
                   "header0_words": [int(w) for w in fh.header0.words]}
# ---- VEGAS and Breakthrough Listen headers (guppi/tests/test_guppi.py:833-853)
with guppi.open(SAMPLE_VEGAS, 'rs') as fh:
    h0 = fh.header0
    out["vegas"] = {"payload_nbytes": h0.payload_nbytes, "bps": h0.bps, "complex_data": bool(h0.complex_data),
                    "npol": h0.npol, "nchan": h0.nchan, "sample_rate_Hz": float(h0.sample_rate.to_value(u.Hz)),
                    "sideband": bool(h0.sideband), "overlap": int(h0.overlap), "offset_s": float(h0.offset.to_value(u.s)),
                    "nbytes": h0.nbytes, "samples_per_frame": int(h0.samples_per_frame), "start_time": fh.start_time.isot}
with guppi.open(SAMPLE_BLC, 'rs') as fh:
    h0 = fh.header0
    out["blc"] = {"nbytes": h0.nbytes, "bps": h0.bps, "complex_data": bool(h0.complex_data), "npol": h0.npol,
                  "nchan": h0.nchan, "samples_per_frame": int(h0.samples_per_frame), "payload_nbytes": h0.payload_nbytes,
                  "sample_rate_Hz": float(h0.sample_rate.to_value(u.Hz)), "start_time": fh.start_time.isot}
with open(os.path.join(GOLD, 'new_edv_cases.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: (v if k not in ('edv4', 'edv42') else v['class']) for k, v in out.items()})[:1800])
