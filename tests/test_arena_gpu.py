"""The output arena (include/bbdecode_arena.h, baseband_amd/arena.py,
placement.py): tensors cut from it are ordinary device tensors that decode
bit-exactly, blocks come back when their tensors die, the allocator grows on
demand, coalesces and trims, and the readers allocate their outputs from it."""
import ctypes as C
import gc

import numpy as np
import pytest

from conftest import golden_path, load_expected, bits_equal

pytestmark = pytest.mark.gpu

GIB = 1 << 30


@pytest.fixture
def small_arena(monkeypatch):
    from baseband_amd import arena
    # (the library backs the arena in steps of 48 GiB by default; small steps here)
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '2')
    ar = arena.Arena(8 * GIB)
    yield ar
    ar.close()


def test_default_growth_step_is_wide(monkeypatch):
    """The first block takes a wide step (48 GiB, or what capacity and free
    memory allow): blocks must span a wide physical range to decode fast
    (profiles/r03k_exp_arena_history.log); trim gives it back."""
    from baseband_amd import arena
    monkeypatch.delenv('BB_ARENA_STEP_GIB', raising=False)
    ar = arena.Arena(64 * GIB)
    try:
        t = ar.empty(1 << 20)
        st = ar.stats()
        assert st['bytes_backed'] == 48 * GIB and st['steps'] == 1
        del t
        gc.collect()
        assert ar.trim() == 48 * GIB
        small = arena.Arena(5 * GIB)
        t = small.empty(1 << 20)
        assert small.stats()['bytes_backed'] == 5 * GIB               # the capacity bounds the step
        del t
        small.close()
    finally:
        ar.close()


def test_arena_tensors_are_plain_tensors_and_decode_right(manifest, small_arena):
    import torch
    from baseband_amd import vdif
    case = manifest['vdif_cfg2_small']
    exp = load_expected('vdif_cfg2_small')
    st = small_arena.stats()
    assert st['capacity'] == 8 * GIB and st['bytes_backed'] == 0 and st['blocks'] == 0
    G = st['chunk_bytes']
    assert G == 32 << 20
    out = small_arena.empty((exp.shape[0],))
    assert out.shape == (exp.shape[0],) and out.dtype == torch.float32 and out.is_cuda
    assert out.data_ptr() % (2 << 20) == 0 and small_arena.owns(out)
    assert small_arena.stats()['bytes_backed'] == 2 * GIB          # grown by the minimum step
    with vdif.open(golden_path(case['file']), 'rs', sample_rate=case['frame_rate'] * case['samples_per_frame']) as fh:
        assert fh.read(out=out) is out
    assert bits_equal(out.cpu().numpy(), exp.reshape(-1))
    # torch ops work on it like on any tensor; views keep the block alive
    assert float((out * 0 + 1).sum()) == out.numel()
    view = out[5:9]
    assert small_arena.stats()['blocks'] == 1
    del out
    gc.collect()
    assert small_arena.stats()['blocks'] == 1
    del view
    gc.collect()
    assert small_arena.stats()['blocks'] == 0 and small_arena.stats()['bytes_in_use'] == 0
    c = small_arena.empty((1000, 8, 16), dtype=torch.complex64)
    assert c.shape == (1000, 8, 16) and c.dtype == torch.complex64
    c.fill_(1 + 2j)
    assert complex(c[3, 2, 1]) == 1 + 2j


def test_blocks_are_cut_first_fit_coalesce_grow_and_trim(small_arena):
    import torch
    G = small_arena.stats()['chunk_bytes']
    f = G // 4                                     # float32 per granule
    a = small_arena.empty(3 * f)                   # 3 granules
    b = small_arena.empty(5 * f)                   # 5 granules
    c = small_arena.empty(f + 1)                   # 1 granule + 4 bytes -> 2 granules
    base = a.data_ptr()
    assert base == small_arena.stats()['base']
    assert b.data_ptr() == base + 3 * G and c.data_ptr() == base + 8 * G
    assert small_arena.stats()['bytes_in_use'] == 10 * G
    del b
    gc.collect()
    d = small_arena.empty(4 * f)                   # fits the hole b left
    assert d.data_ptr() == base + 3 * G
    e = small_arena.empty(2 * f)                   # the granule left of the hole is too small
    assert e.data_ptr() == base + 10 * G
    # growth: 3 GiB does not fit behind the 12 granules in use of the first 2 GiB step:
    # a new step that could hold the whole block, at addresses of its own -- here right
    # behind the first step (nothing was burnt in between), so the free tail of
    # the first step and the new step are one range and the block starts in the tail
    big = small_arena.empty(3 * GIB // 4)
    st = small_arena.stats()
    assert big is not None and big.data_ptr() == base + 12 * G
    assert st['steps'] == 2 and st['bytes_backed'] == 5 * GIB
    assert st['va_used'] == 5 * GIB and st['va_reserved'] >= 8 * GIB
    big.fill_(2.5)
    assert float(big[-1]) == 2.5 and float(big[big.numel() // 2]) == 2.5     # across the step boundary
    # more than the capacity allows next to what is backed: None, nothing changes
    assert small_arena.empty(4 * GIB // 4) is None
    assert small_arena.stats()['bytes_backed'] == 5 * GIB
    del a, c, d, e, big
    gc.collect()
    st = small_arena.stats()
    assert st['bytes_in_use'] == 0 and st['largest_free'] == 5 * GIB          # everything coalesced (adjacent steps)
    free0, _ = torch.cuda.mem_get_info()
    assert small_arena.trim() == 5 * GIB
    st = small_arena.stats()
    assert st['bytes_backed'] == 0 and st['steps'] == 0 and st['bytes_trimmed'] == 5 * GIB
    free1, _ = torch.cuda.mem_get_info()
    assert free1 - free0 >= 4 * GIB                                           # the device has it back
    # after a trim the arena grows at NEW addresses (never where memory was unmapped
    # before: csrc/bb_arena.inc) and what is decoded there is right
    x = small_arena.empty(2 * GIB // 4)               # the whole of a new 2 GiB step
    assert x.data_ptr() == base + 5 * GIB and small_arena.stats()['va_used'] == 7 * GIB
    # a live block pins its step; free steps go back whatever their place
    y = small_arena.empty(3 * GIB // 4)
    assert small_arena.stats()['steps'] == 2
    del x
    gc.collect()
    assert small_arena.trim() == 2 * GIB and small_arena.stats()['steps'] == 1       # the FIRST of the two steps
    y.fill_(1.25)
    assert float(y[-1]) == 1.25
    del y
    gc.collect()
    assert small_arena.trim() == 3 * GIB and small_arena.stats()['bytes_backed'] == 0


def test_regrown_memory_holds_what_is_decoded_into_it(monkeypatch):
    """Round 4: trim and grow again at once, several times, decoding DIFFERENT
    data each time into the block and verifying all of it on the device and a
    piece on the host.  With steps mapped at the addresses that had just been
    unmapped, whole 32 MiB granules came back stale or zeroed (the kernel's
    stores went to the pages mapped there before); addresses are never reused
    now.  Also with probing (steps wide enough for the probe launch)."""
    import torch
    from baseband_amd import arena, kernels, _lib
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '10')
    monkeypatch.setenv('BB_ARENA_MIN_GBPS', '7999')          # every candidate counts as slow: all tries are used
    monkeypatch.setenv('BB_ARENA_TRIES', '3')
    dev = torch.device('cuda')
    npol, nchan, blk = 2, 64, 64 << 20
    T = blk // (npol * nchan * 2)
    nfr = 8
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    ar = arena.Arena(32 * GIB)
    try:
        ptrs = set()
        for rnd in range(5):
            image = torch.randint(0, 256, (nfr * blk,), generator=g, device=dev, dtype=torch.uint8)
            b = image.view(torch.int8).view(nfr, nchan, T, npol, 2)
            keep = T - 64 * rnd                      # another row layout every round
            o = ar.empty(nfr * keep * npol * nchan * 2)
            ptrs.add(o.data_ptr())
            kernels.decode_i8_tiled(image, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, keep, src0=0, src_stride=blk, out=o)
            torch.cuda.synchronize()
            ov = o.view(nfr, keep, npol, nchan, 2)
            for fr in range(nfr):
                assert torch.equal(ov[fr], b[fr, :, :keep].permute(1, 2, 0, 3).to(torch.float32)), (rnd, fr)
            host = ov[nfr - 1, :2048].cpu().numpy()
            assert np.array_equal(host, b[nfr - 1, :, :2048].permute(1, 2, 0, 3).to(torch.float32).cpu().numpy())
            st = ar.stats()
            assert st['steps'] == 1 and 2 * (rnd + 1) <= st['probes'] <= 3 * (rnd + 1)     # (two within 4 %: no third)
            del o, ov
            gc.collect()
            assert ar.trim() == 10 * GIB
        assert len(ptrs) == 5, "a step was mapped at addresses used before"
        assert ar.stats()['va_used'] >= 5 * 3 * 10 * GIB          # two or three candidates + the best mapped again, per growth
    finally:
        ar.close()


def test_a_used_up_virtual_range_is_followed_by_another(monkeypatch):
    """Round 5: when the bump pointer reaches the
    end of the reserved range, ANOTHER range is reserved -- it was BB_ERANGE and
    plain allocations for the rest of the process before.  The used-up range
    STAYS reserved: were it given back, the next reservation could return its
    addresses, and memory mapped where a mapping was before reads back wrong
    (tools/experiments/va_reuse_probe.cpp, profiles/r05e_va_reuse.log).  Blocks
    of the old range stay valid and are freed normally."""
    import torch
    from baseband_amd import arena
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '2')
    monkeypatch.setenv('BB_ARENA_VA_GIB', '6')
    ar = arena.Arena(12 * GIB)
    try:
        st = ar.stats()
        assert st['va_reserved'] == 6 * GIB and st['va_ranges'] == 1 and st['va_ranges_made'] == 1
        bases = {st['base']}
        for k in range(3):
            t = ar.empty(1 << 20)
            assert t is not None
            t.fill_(k)
            del t
            gc.collect()
            assert ar.trim() == 2 * GIB
        assert ar.stats()['va_used'] == 6 * GIB and ar.stats()['va_ranges'] == 1
        keep = ar.empty(1 << 20)                    # the range is used up: this one lies in a new one
        assert keep is not None and ar.owns(keep)
        keep.fill_(7.)
        st = ar.stats()
        assert st['va_ranges'] == 2 and st['va_ranges_made'] == 2 and st['base'] not in bases
        bases.add(st['base'])
        others = []
        for k in range(3):                          # (the first fits behind `keep`; two more steps fill the range)
            others.append(ar.empty((2 * GIB - (64 << 20)) // 4))
            assert others[-1] is not None
        assert ar.stats()['va_ranges_made'] == 2 and ar.stats()['steps'] == 3
        more = ar.empty((3 * GIB) // 2 // 4)        # 1.5 GiB: no room in the rests of the live steps -> a step in range 3
        st = ar.stats()
        assert more is not None and st['va_ranges_made'] == 3 and st['va_ranges'] == 3 and st['base'] not in bases
        assert ar.owns(keep) and ar.owns(more) and float(keep.sum()) == 7. * keep.numel()
        del keep, others
        gc.collect()
        assert ar.trim() == 6 * GIB
        st = ar.stats()
        assert st['va_ranges'] == 3 and st['blocks'] == 1 and st['steps'] == 1
        more.fill_(2.)
        assert float(more[::1024].sum()) == 2. * more[::1024].numel()
        assert not ar.owns(torch.empty(4, device='cuda'))
    finally:
        ar.close()


def test_a_thousand_grow_and_trim_cycles_still_serve_blocks(monkeypatch):
    """A service that idles between bursts trims and regrows for ever: 1,000
    cycles through 4 GiB virtual ranges in 1 GiB steps (250 ranges, all kept
    reserved) -- every block is served by the arena, at an address never used
    before, and holds what is written into it."""
    import torch
    from baseband_amd import arena
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '1')
    monkeypatch.setenv('BB_ARENA_VA_GIB', '4')
    ar = arena.Arena(2 * GIB)
    try:
        seen = set()
        for k in range(1000):
            t = ar.empty((96 << 20) // 4)
            assert t is not None and ar.owns(t), k
            if k % 10 == 0:                      # (a mapping at a reused address reads back wrong: it did, r05d)
                t.fill_(float(k))
                assert float(t[::1024].sum()) == float(k) * t[::1024].numel(), k
            seen.add(t.data_ptr())
            del t
            gc.collect()
            assert ar.trim() == 1 * GIB, k
        st = ar.stats()
        assert st['va_ranges_made'] == 250 and st['va_ranges'] == 250 and st['va_used'] == 1000 * GIB
        assert st['bytes_grown'] == st['bytes_trimmed'] == 1000 * GIB and st['bytes_backed'] == 0
        assert len(seen) == 1000                 # no address ever came back (used-up ranges stay reserved)
    finally:
        ar.close()


def test_a_growth_that_fails_burns_no_addresses(monkeypatch):
    """addresses are taken only once a step's memory exists -- a
    growth that cannot get device memory leaves the range as it was."""
    import torch
    from baseband_amd import arena
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '2')
    free_b, total_b = torch.cuda.mem_get_info()
    ar = arena.Arena(2 * total_b)
    try:
        for _ in range(3):
            assert ar.empty(int(free_b * 1.5) // 4) is None       # more than the device has
        assert ar.stats()['va_used'] == 0 and ar.stats()['bytes_backed'] == 0
        t = ar.empty(1 << 20)
        assert t is not None and ar.stats()['va_used'] == 2 * GIB
    finally:
        ar.close()


def test_prepare_grows_in_the_background_and_alloc_waits_for_it(monkeypatch):
    """bb_arena_prepare: returns at once, the step arrives
    on a thread of the library, the next block comes out of it (no second
    step), kernels on the caller's streams run meanwhile; a prepare that is not
    needed starts nothing."""
    import time
    import torch
    from baseband_amd import arena
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '16')
    ar = arena.Arena(64 * GIB)
    try:
        t0 = time.perf_counter()
        assert ar.prepare(3 * GIB)
        dt_call = time.perf_counter() - t0
        st = ar.stats()
        assert st['prepares'] == 1
        x = torch.ones(1 << 20, device='cuda')
        for _ in range(20):                        # the caller's own work goes on
            x = x * 1.0001
        torch.cuda.synchronize()
        t = ar.empty((3 * GIB) // 4)                # waits for the growth in flight if need be
        st = ar.stats()
        assert t is not None and ar.owns(t)
        assert st['steps'] == 1 and st['bytes_backed'] == 16 * GIB and st['growing'] == 0
        # (background steps are probed too since round 6: the second-chance rule needs the figure)
        assert st['probes'] == 1 and st['first_probe_gbps'] > 1000 and st['last_probe_gbps'] == st['first_probe_gbps']
        assert st['prepare_ms'] > 0 and dt_call < max(0.05, st['prepare_ms'] * 1e-3)
        t.fill_(3.)
        assert float(t[::65536].sum()) == 3. * t[::65536].numel()
        assert ar.prepare(1 * GIB) and ar.stats()['prepares'] == 1          # room already: nothing started
        assert ar.prepare(15 * GIB) and ar.stats()['prepares'] == 2         # no room for that: a second step
        u = ar.empty((15 * GIB) // 4)
        assert u is not None and ar.stats()['steps'] == 2
        del t, u
        gc.collect()
        assert ar.trim() == 32 * GIB
        # closing with a growth in flight waits for it and gives everything back
        assert ar.prepare(2 * GIB)
    finally:
        ar.close()
    free_b, total_b = torch.cuda.mem_get_info()
    assert free_b > total_b - 40 * GIB


@pytest.mark.parametrize('background', [False, True])
def test_a_slow_cheap_first_step_gets_one_second_chance(background, monkeypatch):
    """Round 6: a growth whose first candidate probes below BB_ARENA_RETRY_BELOW_GBPS and was
    cheap to create tries ONE more candidate and keeps the faster; the other goes back to
    the device.  Forced here (every rate is 'slow', every creation 'cheap'); switched off
    again the growth takes its first candidate.  Both growth paths; the memory kept must
    hold what is written to it."""
    import torch
    from baseband_amd import arena
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '12')
    monkeypatch.setenv('BB_ARENA_RETRY_BELOW_GBPS', '8000')
    monkeypatch.setenv('BB_ARENA_CHEAP_MS_PER_GIB', '100000')
    monkeypatch.delenv('BB_ARENA_MAX_CANDIDATES', raising=False)
    free0, _ = torch.cuda.mem_get_info()
    ar = arena.Arena(64 * GIB)
    try:
        if background:
            assert ar.prepare(3 * GIB)
        t = ar.empty((3 * GIB) // 4)
        st = ar.stats()
        # (every rate counts as slow here: all three candidates are looked at, the fastest stays)
        assert st['second_chances'] == 1 and st['probes'] == 3 and st['second_chance_wins'] in (0, 1)
        assert st['steps'] == 1 and st['bytes_backed'] == 12 * GIB          # the losers are gone
        assert st['last_probe_gbps'] >= st['first_probe_gbps'] > 1000
        assert (st['last_probe_gbps'] > st['first_probe_gbps']) == bool(st['second_chance_wins'])
        assert len(st['probe_history']) == 3 and abs(max(st['probe_history']) - st['last_probe_gbps']) <= 1
        assert abs(st['probe_history'][0] - st['first_probe_gbps']) <= 1
        free1, _ = torch.cuda.mem_get_info()
        assert free0 - free1 < 14 * GIB
        t.fill_(5.)
        torch.cuda.synchronize()
        assert float(t[::4099].min()) == 5. and float(t[-1]) == 5.
        # the rule switched off: the next step is taken as it comes
        monkeypatch.setenv('BB_ARENA_RETRY_BELOW_GBPS', '0')
        u = ar.empty((11 * GIB) // 4)
        st = ar.stats()
        assert u is not None and st['steps'] == 2 and st['second_chances'] == 1 and st['probes'] == 4
        assert len(st['probe_history']) == 4
        u.fill_(6.)
        torch.cuda.synchronize()
        assert float(u[::4099].max()) == 6. and float(t[::4099].max()) == 5.
        del t, u
    finally:
        ar.close()


def test_opening_a_large_stream_prepares_the_arena(tmp_path, monkeypatch):
    """open() of a stream that decodes to >= 1 GiB starts the arena's first step
    in the background; BB_ARENA_PREPARE=0 does not; a small stream never does."""
    import torch
    from baseband_amd import vdif, arena, placement, synth
    monkeypatch.setenv('BB_ARENA_STEP_GIB', '4')
    image, h0 = synth.random_vdif(5, 9000, payload_nbytes=8000, frame_rate=1000)       # 72 MB -> 1.15 GB decoded
    path = tmp_path / 'big.vdif'
    path.write_bytes(image.tobytes())
    small, _ = synth.random_vdif(6, 100, payload_nbytes=8000, frame_rate=1000)
    spath = tmp_path / 'small.vdif'
    spath.write_bytes(small.tobytes())
    arena.disable()
    try:
        with vdif.open(str(spath), 'rs', sample_rate=32e6) as fh:
            assert arena.default() is None                                            # nothing to prepare for
        monkeypatch.setenv('BB_ARENA_PREPARE', '0')
        with vdif.open(str(path), 'rs', sample_rate=32e6) as fh:
            assert arena.default() is None
        monkeypatch.delenv('BB_ARENA_PREPARE')
        with vdif.open(str(path), 'rs', sample_rate=32e6) as fh:
            ar = arena.default()
            assert ar is not None and ar.stats()['prepares'] == 1
            got = fh.read()
            st = ar.stats()
            assert ar.owns(got) and st['steps'] == 1 and st['bytes_backed'] == 4 * GIB
        exp = kernels_reference(image, h0)
        assert torch.equal(got.view(torch.int32).cpu(), torch.from_numpy(exp.view(np.int32)))
    finally:
        del got
        gc.collect()
        arena.disable()


def kernels_reference(image, h0):
    """2-bit VDIF payloads re-expanded on the host from the library's own level table."""
    from baseband_amd import _lib
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    pay = image.reshape(-1, 8032)[:, 32:].reshape(-1)
    return lev[(pay[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3].reshape(-1)


def test_raw_abi_rejects_what_is_not_a_block(small_arena):
    from baseband_amd import _lib
    lib, h = _lib.lib, small_arena._handle
    G = small_arena.stats()['chunk_bytes']
    p = C.c_void_p()
    assert lib.bb_arena_alloc(h, 100, C.byref(p)) == _lib.BB_OK and p.value
    q = C.c_void_p()
    assert lib.bb_arena_alloc(h, 100, C.byref(q)) == _lib.BB_OK and q.value == p.value + G
    assert lib.bb_arena_free(h, C.c_void_p(p.value + 2 * G)) == _lib.BB_EINVAL    # not a live block
    assert lib.bb_arena_free(h, C.c_void_p(p.value + 8)) == _lib.BB_EINVAL        # not a block start
    assert lib.bb_arena_free(h, C.c_void_p(p.value)) == _lib.BB_OK
    assert lib.bb_arena_free(h, C.c_void_p(p.value)) == _lib.BB_EINVAL            # twice
    assert lib.bb_arena_free(h, C.c_void_p(q.value)) == _lib.BB_OK
    assert lib.bb_arena_create(0, C.byref(p)) == _lib.BB_EINVAL
    # a capacity beyond the device's memory is only a virtual range; blocks fail when memory runs out
    huge = C.c_void_p()
    assert lib.bb_arena_create(1 << 40, C.byref(huge)) == _lib.BB_OK
    assert lib.bb_arena_alloc(huge, 1 << 39, C.byref(p)) == _lib.BB_ERANGE and not p.value
    assert lib.bb_arena_destroy(huge) == _lib.BB_OK


def test_readers_take_their_outputs_from_the_arena(manifest, monkeypatch):
    """placement.empty_output: read() results of at least ARENA_MIN_BYTES live
    in the process-wide arena (created on first use) and are freed into it;
    smaller ones, and everything when the arena is switched off, are torch
    allocations.  (BB_ARENA_KEEP=1: no automatic trim, so that the explicit
    `release_unused` can be checked; the automatic one has its own test.)"""
    monkeypatch.setenv('BB_ARENA', '1')
    monkeypatch.setenv('BB_ARENA_KEEP', '1')
    import torch
    import baseband_amd
    from baseband_amd import arena, placement, vdif
    case = manifest['vdif_cfg2_small']
    exp = load_expected('vdif_cfg2_small')
    kw = dict(sample_rate=case['frame_rate'] * case['samples_per_frame'])
    arena.disable()
    try:
        monkeypatch.setattr(placement, 'ARENA_MIN_BYTES', 1 << 16)
        with vdif.open(golden_path(case['file']), 'rs', **kw) as fh:
            fh.decode_ahead = False
            got = fh.read()
        ar = arena.default()
        assert ar is not None, "the first output did not create the arena"
        assert ar.owns(got), "read() output is not in the arena"
        assert bits_equal(got.cpu().numpy(), exp.reshape(got.shape))
        assert ar.stats()['blocks'] >= 1
        del got
        gc.collect()
        assert ar.stats()['blocks'] == 0
        mine = baseband_amd.empty_output((4 << 20,))
        assert ar.owns(mine) and not ar.owns(torch.empty(4, device='cuda'))
        small = baseband_amd.empty_output((100,))
        assert not ar.owns(small)                                  # below the threshold: torch's allocator
        del mine
        gc.collect()
        assert placement.release_unused() == ar.stats()['bytes_trimmed'] > 0
        assert ar.stats()['bytes_backed'] == 0
    finally:
        arena.disable()
    monkeypatch.setenv('BB_ARENA', '0')
    with vdif.open(golden_path(case['file']), 'rs', **kw) as fh:
        got = fh.read()
    assert bits_equal(got.cpu().numpy(), exp.reshape(got.shape))
    assert arena.default() is None


def test_large_read_lifecycle_through_the_default_arena(tmp_path, monkeypatch):
    """A 600 MB VDIF file read whole (9.6 GB of output) with the defaults a
    user gets: the output lives in the arena, whose first step was PROBED; the
    samples are right; deleting the result frees the block; `release_unused`
    gives the memory back to the device; threads may allocate concurrently."""
    monkeypatch.setenv('BB_ARENA', '1')
    monkeypatch.setenv('BB_ARENA_KEEP', '1')
    import threading
    import torch
    from baseband_amd import arena, placement, synth, vdif
    import bb_oracle_np as orc
    monkeypatch.delenv('BB_ARENA_GIB', raising=False)
    monkeypatch.delenv('BB_ARENA_STEP_GIB', raising=False)
    arena.disable()
    nframes = 75000
    image, h0 = synth.random_vdif(77, nframes, payload_nbytes=8000, frame_rate=1000)
    path = tmp_path / 'big.vdif'
    image.tofile(str(path))
    try:
        free0, _ = torch.cuda.mem_get_info()
        with vdif.open(str(path), 'rs', sample_rate=32e6) as fh:
            got = fh.read()
        ar = arena.default()
        assert ar is not None and ar.owns(got)
        st = ar.stats()
        # (the step was started by open(): placement.prepare_output -> bb_arena_prepare, whose
        # background steps are not probed; a step the read itself grows is)
        assert st['steps'] == 1 and (st['prepares'] == 1 or (st['probes'] >= 1 and st['last_probe_gbps'] > 3000)), st
        assert st['bytes_backed'] >= 9 << 30
        # spot check against the oracle: first, a middle and the last frame
        for f in (0, nframes // 2, nframes - 1):
            want, _ = orc.vdif_read(image[f * 8032:(f + 1) * 8032], frame_rate=1000)
            assert bits_equal(got[f * 32000:(f + 1) * 32000].cpu().numpy(), want.reshape(-1))
        del got
        gc.collect()
        assert ar.stats()['blocks'] == 0
        # a loop of small reads (and its 64 MiB read-ahead window) stays out of the arena
        with vdif.open(str(path), 'rs', sample_rate=32e6) as fh:
            for _ in range(8):
                piece = fh.read(32000)
            assert fh._decoded is not None and not ar.owns(fh._decoded[2]) and not ar.owns(piece)
        assert ar.stats()['blocks'] == 0
        # concurrent allocations from two threads
        monkeypatch.setattr(placement, 'ARENA_MIN_BYTES', 64 << 20)
        errs = []

        def worker(seed):
            try:
                for k in range(20):
                    t = placement.empty_output((64 << 20) // 4 * (1 + (seed + k) % 3))
                    t[:4] = seed
                    assert ar.owns(t) and float(t[0]) == seed
                    del t
            except Exception as exc:        # pragma: no cover
                errs.append(exc)
        ths = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not errs, errs
        gc.collect()
        assert ar.stats()['blocks'] == 0
        assert placement.release_unused() == st['bytes_backed']
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 >= free0 - (2 << 30), (free0, free1)
    finally:
        arena.disable()


def test_blocks_reused_on_another_stream_wait_for_the_work_left_behind():
    """A block freed while work is still queued on it (the readers return
    before their decode is done) and handed out again on ANOTHER stream: that
    stream is made to wait for what had been queued at the free, so the new
    owner's writes land after the old owner's."""
    import torch
    from baseband_amd import arena
    ar = arena.Arena(4 << 30)
    try:
        n = (512 << 20) // 4
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        big = torch.empty(256 << 20, dtype=torch.float32, device='cuda')
        for rep in range(5):
            with torch.cuda.stream(s1):
                a = ar.empty(n)
                p = a.data_ptr()
                for _ in range(20):                 # a long queue of work, then writes into the block
                    big.mul_(1.0001)
                a.fill_(1.0)
                del a                               # freed at once, the fills are still queued
            with torch.cuda.stream(s2):
                b = ar.empty(n)
                assert b.data_ptr() == p            # first fit: the same memory
                b.fill_(2.0)
            torch.cuda.synchronize()
            assert float(b.min()) == 2.0 and float(b.max()) == 2.0, rep
            del b
        # same stream: no events needed, and the bookkeeping stays bounded
        for _ in range(200):
            t = ar.empty(n)
            del t
        assert len(ar._freed) <= 65
    finally:
        ar.close()


def test_trim_waits_for_work_queued_on_freed_blocks():
    """a block is freed at garbage collection while its
    decode may still be queued on a side stream; `trim()` / `close()` unmap the
    memory, so they first wait for the events recorded at each free.  Here a
    long chain of fills is queued on a side stream, the tensor dropped and the
    arena trimmed at once: no fault, the device is alive afterwards, the memory
    is back."""
    import torch
    from baseband_amd import arena
    ar = arena.Arena(8 << 30)
    try:
        side = torch.cuda.Stream()
        n = (1 << 30) // 4
        with torch.cuda.stream(side):
            t = ar.empty(n)
            for k in range(40):                       # tens of ms of queued writes into the block
                t.fill_(float(k))
            del t                                     # freed with the work still in flight
        gc.collect()
        assert ar.stats()['blocks'] == 0 and ar.stats()['bytes_backed'] > 0
        released = ar.trim()                          # must not unmap under the running fills
        assert released > 0 and ar.stats()['bytes_backed'] == 0
        assert side.query(), "trim() returned before the work on the freed block had finished"
        torch.cuda.synchronize()
        assert float(torch.ones(4, device='cuda').sum()) == 4.0
        # the same through close()
        with torch.cuda.stream(side):
            t = ar.empty(n)
            for k in range(40):
                t.fill_(float(k))
            del t
        gc.collect()
    finally:
        ar.close()
    torch.cuda.synchronize()
    assert float(torch.ones(4, device='cuda').sum()) == 4.0


def test_unsupported_dtype_takes_no_block():
    """a dtype the array interface table does not know is
    refused BEFORE a block is taken (None -> the caller uses torch.empty), and
    `empty_output` falls back to torch for it."""
    import torch
    import baseband_amd
    from baseband_amd import arena, placement
    ar = arena.Arena(4 << 30)
    try:
        for dt in (torch.float16, torch.bfloat16, torch.int16, torch.bool, torch.complex128):
            assert ar.empty(1 << 20, dtype=dt) is None
        assert ar.stats()['blocks'] == 0 and ar.stats()['bytes_in_use'] == 0
        assert ar.empty(1 << 20, dtype=torch.float64) is not None
    finally:
        ar.close()
    t = baseband_amd.empty_output(((1 << 30) // 2 + 8,), dtype=torch.float16)       # >= 1 GiB
    assert t.dtype == torch.float16 and t.is_cuda
    ar = arena.default()
    assert ar is None or not ar.owns(t)


def test_one_arena_per_device_created_once_under_a_lock(monkeypatch):
    """VERDICT r3 next 4c / ADVICE r3 (low): the readers' arenas are kept per
    device; lazy creation from several threads at once yields ONE arena and
    never replaces (closes) one that exists; only an explicit enable() does."""
    monkeypatch.setenv('BB_ARENA', '1')
    import threading
    import torch
    from baseband_amd import arena, placement
    arena.disable()
    try:
        dev = torch.device('cuda', 0)
        got = []
        ths = [threading.Thread(target=lambda: got.append(placement._arena_for(dev))) for _ in range(8)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert len(got) == 8 and all(g is got[0] and g is not None for g in got)
        assert arena.default(dev) is got[0] and arena.default() is got[0]
        assert arena.default(torch.device('cuda', 7)) is None           # (no arena there: never another device's)
        assert got[0].device == dev
        t = got[0].empty(1 << 20)
        assert placement._arena_for(dev) is got[0] and got[0].owns(t)   # still the same, tensor still valid
        t.fill_(3.0)
        assert float(t[5]) == 3.0
        new = arena.enable(2 << 30, device=0)                           # the program's own choice replaces it
        assert arena.default(dev) is new and new is not got[0]
    finally:
        arena.disable()


def test_arena_gives_memory_back_when_nothing_needs_it(tmp_path, monkeypatch):
    """the arena trims by itself when its last block dies
    and no reader is open; while a reader is open the memory stays (the next
    read reuses it); a step is at most half of the free memory."""
    monkeypatch.setenv('BB_ARENA', '1')
    monkeypatch.setenv('BB_ARENA_IDLE_S', '0')           # (the default waits 30 s; its timer has a test of its own)
    monkeypatch.delenv('BB_ARENA_KEEP', raising=False)
    monkeypatch.delenv('BB_ARENA_GIB', raising=False)
    import torch
    from baseband_amd import arena, placement, synth, vdif
    arena.disable()
    nframes = 40000                                  # 5.1 GB of output
    image, h0 = synth.random_vdif(78, nframes, payload_nbytes=8000, frame_rate=1000)
    path = tmp_path / 'mid.vdif'
    image.tofile(str(path))
    try:
        torch.cuda.empty_cache()
        free0, _ = torch.cuda.mem_get_info()
        fh = vdif.open(str(path), 'rs', sample_rate=32e6)
        got = fh.read()
        ar = arena.default()
        assert ar is not None and ar.owns(got)
        backed = ar.stats()['bytes_backed']
        assert 0 < backed <= free0 // 2, (backed, free0)
        del got
        gc.collect()
        assert ar.stats()['bytes_backed'] == backed, "trimmed although a reader is open"
        fh.seek(0)
        got = fh.read()                              # served from the same step
        assert ar.stats()['bytes_grown'] == backed
        fh.close()
        assert ar.stats()['bytes_backed'] == backed, "trimmed although a block is alive"
        last = got[-32000:].cpu().numpy()
        del got
        gc.collect()
        assert ar.stats()['bytes_backed'] == 0, "the last block died with no reader open: memory goes back"
        import bb_oracle_np as orc
        want, _ = orc.vdif_read(image[(nframes - 1) * 8032:], frame_rate=1000)
        assert bits_equal(last, want.reshape(-1))
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 >= free0 - (2 << 30), (free0, free1)
    finally:
        arena.disable()


def test_two_readers_and_a_callers_allocation_of_the_rest(tmp_path, monkeypatch):
    """VERDICT r3 next 4 "done" test: with the arena limited to 64 GiB
    (BB_ARENA_GIB), two readers each return an output of 1 GiB or more, and a
    caller's own torch.empty of the REST of the device (what was free at the
    start minus 64 GiB and a margin) does not run out of memory while both
    outputs are alive: growing probes at most two steps at a time and gives
    the slower one back at once, and never holds more than the capacity."""
    monkeypatch.setenv('BB_ARENA', '1')
    monkeypatch.setenv('BB_ARENA_GIB', '64')
    monkeypatch.setenv('BB_ARENA_IDLE_S', '0')
    monkeypatch.delenv('BB_ARENA_KEEP', raising=False)
    import torch
    from baseband_amd import arena, synth, vdif
    arena.disable()
    nframes = 12000                                  # 1.5 GB of output each
    paths = []
    for k in range(2):
        image, h0 = synth.random_vdif(80 + k, nframes, payload_nbytes=8000, frame_rate=1000)
        p = tmp_path / 'r{}.vdif'.format(k)
        image.tofile(str(p))
        paths.append(str(p))
    try:
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        free0, total = torch.cuda.mem_get_info()
        fhs = [vdif.open(p, 'rs', sample_rate=32e6) for p in paths]
        outs = [fh.read() for fh in fhs]
        ar = arena.default()
        assert ar is not None and all(ar.owns(o) for o in outs)
        st = ar.stats()
        assert st['capacity'] == 64 << 30 and st['bytes_backed'] <= 64 << 30
        rest = free0 - (64 << 30) - (6 << 30)
        assert rest > 0
        mine = torch.empty(rest, dtype=torch.uint8, device='cuda')      # must not raise OutOfMemoryError
        mine[:16] = 1
        mine[-16:] = 2
        torch.cuda.synchronize()
        assert int(mine[0]) == 1 and int(mine[-1]) == 2
        more = [fh.read(32000 * 100) for fh in fhs if not fh.seek(0)]   # the readers still work next to it
        assert all(m.numel() == 3200000 for m in more)
        del mine, more
        for fh in fhs:
            fh.close()
        del outs
        gc.collect()
        assert ar.stats()['bytes_backed'] == 0
    finally:
        arena.disable()
        torch.cuda.empty_cache()


def test_idle_arena_trims_after_the_delay_and_not_between_reads(monkeypatch):
    """The automatic trim waits BB_ARENA_IDLE_S with nothing alive: a block
    taken inside the delay cancels it (no trim / regrow between two reads of a
    loop), an arena left alone gives its memory back."""
    import time
    import torch
    import baseband_amd
    from baseband_amd import arena, placement
    monkeypatch.setenv('BB_ARENA', '1')
    monkeypatch.setenv('BB_ARENA_IDLE_S', '1.0')
    monkeypatch.delenv('BB_ARENA_KEEP', raising=False)
    monkeypatch.setattr(placement, '_idle_trims', 0)         # (every idle trim doubles the next delay)
    arena.disable()
    try:
        monkeypatch.setattr(placement, 'ARENA_MIN_BYTES', 1 << 20)
        t = baseband_amd.empty_output((1 << 22,))
        ar = arena.default()
        assert ar is not None and ar.owns(t)
        backed = ar.stats()['bytes_backed']
        for k in range(3):                                   # "reads" 0.4 s apart: never trimmed
            del t
            gc.collect()
            time.sleep(0.4)
            assert ar.stats()['bytes_backed'] == backed
            t = baseband_amd.empty_output((1 << 22,))
            assert ar.stats()['bytes_grown'] == backed
        time.sleep(1.5)
        assert ar.stats()['bytes_backed'] == backed, "trimmed under a live block"
        del t
        gc.collect()
        assert ar.stats()['bytes_backed'] == backed          # not at once ...
        time.sleep(2.0)
        assert ar.stats()['bytes_backed'] == 0               # ... but after the delay
        # the next idle trim waits twice as long: trim + regrow burns addresses (placement._idle_seconds)
        assert placement._idle_trims == 1 and placement._idle_seconds() == 2.0
        t = baseband_amd.empty_output((1 << 22,))
        del t
        gc.collect()
        time.sleep(1.4)
        assert ar.stats()['bytes_backed'] == backed          # 1.0 s would have trimmed by now
        time.sleep(1.6)
        assert ar.stats()['bytes_backed'] == 0 and placement._idle_trims == 2
    finally:
        arena.disable()
