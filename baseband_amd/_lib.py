"""ctypes binding of libbbdecode.so (the C ABI declared in include/bbdecode.h).

The library is the only compute path of this package: if it cannot be loaded
the import fails loudly -- there is no NumPy/PyTorch fallback.
"""
import ctypes as C
import os

# PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Import it
# BEFORE loading libbbdecode.so so that the dynamic loader resolves our HIP
# dependency (same SONAME) to the runtime torch already initialised: two HIP
# runtimes in one process do not see each other's devices, streams or memory.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# BB_EXPERIMENTS=1 in the environment loads the experiment build (make -C
# baseband_amd/csrc EXPERIMENTS=1): the same kernels plus the measurement
# variants and knobs of include/bbdecode_exp.h.  tools/experiments/exp_*.py need it; the
# package, the tests and bench.py run on the product library.
EXPERIMENTS = os.environ.get('BB_EXPERIMENTS', '') not in ('', '0')
LIB_PATH = os.path.join(_HERE, 'libbbdecode_exp.so' if EXPERIMENTS else 'libbbdecode.so')

BB_OK = 0
BB_EIO = -5
BB_EINVAL = -22
BB_ERANGE = -34
BB_ENOTSUP = -95

CODER_VDIF = 0
CODER_MARK5B = 1
CODER_INT = 2

FRAME_OK = 0x1
FRAME_INVALID = 0x2

# include/bbdecode_tune.h (geometry knobs of the product library)
TUNE_BLOCKS = 2
TUNE_TILE_ELEMS = 4
TUNE_ENCODE_DIRECT = 5
TUNE_GATHER_BYTES = 6
TUNE_TILES_PER_WAVE = 7
TUNE_TILED_STAGE = 10
TUNE_MKBF_CHANNELS = 11
TUNE_GATHER_CHUNKS = 12
TUNE_SEG_TILES = 13
TUNE_WORK_STRIPES = 18
TUNE_XPOSE = 19
TUNE_XPOSE_ROWS = 20
TUNE_M4_WIDEN = 22
TUNE_SELECT_BYTES = 23
TUNE_LUT_TILES = 24
TUNE_M4_TILES = 26
TUNE_XPOSE_TC = 27
TUNE_XPOSE_MIN_NC = 28
TUNE_ENCODE_RUNS = 30
TUNE_GATHER_GLDS = 36
TUNE_ENCODE_STRIPES = 38
TUNE_SELECT_PICK = 39
TUNE_PICK_BYTES = 40
TUNE_VDIF8_LDS_GIB = 41
TUNE_TOUCH_MIB = 42
# include/bbdecode_exp.h (experiment build only: bb_tune answers BB_EINVAL otherwise)
TUNE_FLAT_VARIANT = 0
TUNE_NT_STORES = 1
TUNE_NT_LOADS = 3
TUNE_TILES_PER_WAVE_8BIT = 8
TUNE_LDS_PAD = 9
TUNE_FRONT_GROUP = 14
TUNE_FRONT_STEPS = 15
TUNE_OUT_STRIPE_W = 16
TUNE_OUT_STRIPE_S = 17
TUNE_BYTE_LUT = 21
TUNE_LUT_SMALL = 25
TUNE_FLAT8_LDS = 29
TUNE_BURST = 31
TUNE_COPY = 35
TUNE_M4_LDS = 37
TUNE_BURST_BYTES = 32
TUNE_BURST_PERIOD = 33
TUNE_BURST_WAVES = 34


class BBError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("{}: {} (code {}, hip error {})".format(
            where, lib.bb_strerror(code).decode(), code, lib.bb_last_hip_error()))


class FrameRec(C.Structure):
    _fields_ = [('payload_offset', C.c_int64), ('time_index', C.c_int32),
                ('thread_id', C.c_int16), ('flags', C.c_uint16)]


class VDIFScanParams(C.Structure):
    _fields_ = [('first_offset', C.c_uint64), ('frame_nbytes', C.c_uint32),
                ('header_nbytes', C.c_uint32), ('pattern', C.c_uint32 * 8),
                ('mask', C.c_uint32 * 8), ('ref_seconds', C.c_int32),
                ('ref_frame_nr', C.c_int32), ('frame_rate', C.c_int32),
                ('set_nframes', C.c_int32)]


class Mark5BScanParams(C.Structure):
    _fields_ = [('first_offset', C.c_uint64), ('ref_seconds', C.c_int32),
                ('ref_frame_nr', C.c_int32), ('frame_rate', C.c_int32),
                ('by_position', C.c_int32)]


class DecodeParams(C.Structure):
    _fields_ = [('coder', C.c_int32), ('bps', C.c_int32), ('chunk', C.c_int32),
                ('nslot', C.c_int32), ('payload_nbytes', C.c_uint64),
                ('src0', C.c_int64), ('src_stride', C.c_int64),
                ('complex_data', C.c_int32), ('fill_re', C.c_float),
                ('fill_im', C.c_float), ('reserved', C.c_int32)]


class Mark4ScanParams(C.Structure):
    _fields_ = [('first_offset', C.c_uint64), ('ntrack', C.c_int32),
                ('ref_year', C.c_int32), ('ref_qms', C.c_int64),
                ('frame_qms', C.c_int32), ('by_position', C.c_int32)]


class Mark4DecodeParams(C.Structure):
    _fields_ = [('ntrack', C.c_int32), ('reserved', C.c_int32),
                ('nwords', C.c_uint64), ('fill_words', C.c_uint64),
                ('src0', C.c_int64), ('src_stride', C.c_int64),
                ('sign_bit', C.c_uint8 * 32), ('mag_bit', C.c_uint8 * 32),
                ('fill', C.c_float), ('reserved2', C.c_int32)]


class TiledParams(C.Structure):
    _fields_ = [('layout', C.c_int32), ('npol', C.c_int32), ('nchan', C.c_int32),
                ('nchan_stored', C.c_int32), ('ntime', C.c_uint64),
                ('t_lo', C.c_uint64), ('t_hi', C.c_uint64),
                ('src0', C.c_int64), ('src_stride', C.c_int64),
                ('fill_re', C.c_float), ('fill_im', C.c_float),
                ('npol_stored', C.c_int32), ('pol_first', C.c_int32), ('d_chan_map', C.c_void_p)]


class ArenaStats(C.Structure):
    _fields_ = [('base', C.c_uint64), ('capacity', C.c_uint64), ('bytes_backed', C.c_uint64),
                ('bytes_in_use', C.c_uint64), ('largest_free', C.c_uint64), ('bytes_grown', C.c_uint64),
                ('bytes_trimmed', C.c_uint64), ('chunk_bytes', C.c_uint32), ('steps', C.c_uint32),
                ('blocks', C.c_uint32), ('probes', C.c_uint32), ('last_probe_gbps', C.c_double), ('create_ms', C.c_double),
                ('grow_ms', C.c_double), ('va_reserved', C.c_uint64), ('va_used', C.c_uint64),
                ('va_ranges', C.c_uint32), ('va_ranges_made', C.c_uint32), ('prepares', C.c_uint32),
                ('growing', C.c_uint32), ('prepare_ms', C.c_double), ('prepare_wait_ms', C.c_double),
                ('first_probe_gbps', C.c_double), ('last_create_ms', C.c_double),
                ('second_chances', C.c_uint32), ('second_chance_wins', C.c_uint32),
                ('second_chances_no_room', C.c_uint32), ('probe_history_n', C.c_uint32),
                ('probe_history', C.c_uint16 * 16)]


LAYOUT_GUPPI_CF = 0
LAYOUT_MKBF = 1
LAYOUT_GUPPI_TF = 2


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "baseband_amd: {} not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C baseband_amd/csrc{}` (needs hipcc, gfx950). There is no "
            "CPU fallback.".format(LIB_PATH, ' EXPERIMENTS=1' if EXPERIMENTS else ''))
    return C.CDLL(LIB_PATH)


lib = _load()

_vp = C.c_void_p
_sz = C.c_size_t

# (name, restype, argtypes) -- must list every symbol of include/bbdecode.h
# include/bbdecode_tune.h and include/bbdecode_arena.h
SIGNATURES = [
    ('bb_abi_version', C.c_int, []),
    ('bb_strerror', C.c_char_p, [C.c_int]),
    ('bb_last_hip_error', C.c_int, []),
    ('bb_last_kernel', C.c_char_p, []),
    ('bb_init', C.c_int, []),
    ('bb_get_levels', C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float), _sz]),
    ('bb_get_encode_thresholds', C.c_int, [C.POINTER(C.c_float)]),
    ('bb_vdif_scan', C.c_int, [_vp, _sz, C.POINTER(VDIFScanParams), _vp, _sz, _vp]),
    ('bb_vdif_locate', C.c_int, [_vp, _sz, C.POINTER(VDIFScanParams), _vp, _sz, _vp, _vp]),
    ('bb_vdif_scan_at', C.c_int, [_vp, _sz, C.POINTER(VDIFScanParams), _vp, _sz, _vp, _vp]),
    ('bb_mark5b_scan', C.c_int, [_vp, _sz, C.POINTER(Mark5BScanParams), _vp, _sz, _vp]),
    ('bb_touch', C.c_int, [_vp, _sz, _vp]),
    ('bb_mark5b_locate', C.c_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    ('bb_mark5b_locate_stream', C.c_int, [_vp, _sz, C.c_uint32, C.c_uint32, _vp, _sz, _vp, _vp]),
    ('bb_mark5b_scan_at', C.c_int, [_vp, _sz, C.POINTER(Mark5BScanParams), _vp, _sz, _vp, _vp]),
    ('bb_verify_records', C.c_int, [_vp, _sz, C.c_int32, C.c_uint32, _sz, _vp, _vp]),
    ('bb_build_index', C.c_int, [_vp, _sz, _vp, C.c_int, _vp, _sz, _vp]),
    ('bb_decode_frames', C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(DecodeParams), _vp, _sz, _vp]),
    ('bb_decode_frames_select', C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(DecodeParams), _vp, C.c_int, _vp, _sz, _vp]),
    ('bb_decode_frames_select_check', C.c_int, [C.POINTER(DecodeParams), C.c_int]),
    ('bb_copy_frames', C.c_int, [_vp, _sz, _sz, C.c_uint64, C.c_int64, C.c_int64, _vp, _sz, _vp]),
    ('bb_fetch_counter', C.c_int, [_vp, _vp, _vp, _vp]),
    ('bb_mark5b_read_window', C.c_int, [_vp, _sz, C.POINTER(Mark5BScanParams), _sz, _sz, C.POINTER(DecodeParams),
                                        _vp, C.c_int, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    ('bb_mark4_read_window', C.c_int, [_vp, _sz, C.POINTER(Mark4ScanParams), _sz, _sz,
                                       C.POINTER(Mark4DecodeParams), C.c_int, _vp, _vp, _vp, _sz, _sz, _vp, _vp,
                                       _vp, _vp]),
    ('bb_vdif_read_window', C.c_int, [_vp, _sz, C.POINTER(VDIFScanParams), _sz, _vp, _sz, C.POINTER(DecodeParams),
                                      _vp, C.c_int, _vp, _vp, _vp, _sz, C.c_uint32, _sz, _vp, _vp, _vp, _vp]),
    ('bb_mark4_scan', C.c_int, [_vp, _sz, C.POINTER(Mark4ScanParams), _vp, _sz, _vp]),
    ('bb_mark4_locate', C.c_int, [_vp, _sz, C.c_int, _vp, _sz, _vp, _vp]),
    ('bb_mark4_header_crc', C.c_int, [_vp, _sz, C.c_int, _vp, C.c_int64, _sz, _vp, _vp]),
    ('bb_mark4_scan_at', C.c_int, [_vp, _sz, C.POINTER(Mark4ScanParams), _vp, _sz, _vp, _vp]),
    ('bb_decode_mark4', C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(Mark4DecodeParams), _vp, _sz, _vp]),
    ('bb_decode_mark4_select', C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(Mark4DecodeParams), C.c_int,
                                         _vp, _sz, _vp]),
    ('bb_decode_i8_tiled', C.c_int, [_vp, _sz, _vp, _sz, C.POINTER(TiledParams), _vp, _sz, _vp]),
    ('bb_encode_flat', C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    ('bb_encode_mark4', C.c_int, [_vp, _sz, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), _vp, _sz, _vp]),
    ('bb_tune', C.c_int, [C.c_int, C.c_int]),
    # include/bbdecode_arena.h
    ('bb_arena_create', C.c_int, [_sz, C.POINTER(_vp)]),
    ('bb_arena_alloc', C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    ('bb_arena_prepare', C.c_int, [_vp, _sz]),
    ('bb_arena_owns', C.c_int, [_vp, _vp]),
    ('bb_arena_free', C.c_int, [_vp, _vp]),
    ('bb_arena_trim', C.c_int, [_vp, C.POINTER(_sz)]),
    ('bb_arena_get_stats', C.c_int, [_vp, C.POINTER(ArenaStats)]),
    ('bb_arena_destroy', C.c_int, [_vp]),
]

# include/bbdecode_exp.h
EXPERIMENT_SIGNATURES = [
    ('bb_debug_trace', C.c_int, [_vp]),
    ('bb_host_register', C.c_int, [_vp, _sz]),
    ('bb_host_unregister', C.c_int, [_vp]),
    ('bb_copy_to_device', C.c_int, [_vp, _vp, _sz, _vp]),
]

for _name, _res, _args in SIGNATURES + (EXPERIMENT_SIGNATURES if EXPERIMENTS else []):
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def check(code, where):
    if code != BB_OK:
        if code == BB_ENOTSUP:
            # the reference surfaces an unknown coder as KeyError from the
            # _decoders dict (base/payload.py:314-315)
            raise KeyError("{}: unsupported coder / bits per sample".format(where))
        raise BBError(code, where)


def last_kernel():
    """Name / template arguments / grid of the decode kernel this thread
    launched last (bb_last_kernel)."""
    return lib.bb_last_kernel().decode()


def get_levels(coder, bps):
    """Host copy of the code -> level table the kernels use."""
    import numpy as np
    n = 1 << bps
    out = np.empty(n, dtype=np.float32)
    check(lib.bb_get_levels(coder, bps, out.ctypes.data_as(C.POINTER(C.c_float)), n),
          'bb_get_levels')
    return out


def encode_thresholds():
    """float32[3]: inputs at which the 2-bit encoder steps to code 1, 2, 3."""
    import numpy as np
    out = np.empty(3, np.float32)
    check(lib.bb_get_encode_thresholds(out.ctypes.data_as(C.POINTER(C.c_float))),
          'bb_get_encode_thresholds')
    return out
