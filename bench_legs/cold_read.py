"""cold_first_read: what the FIRST large read of a process costs.

A fresh child process -- started before the parent touches the GPU -- writes a
2 GiB cfg2 VDIF file (page cache warm), loads the library's kernels with a
small read, then times

    cold   open(path).read() + sync: the first read that needs the output arena
           (34 GB of output: the arena takes its first step; pinned staging
           buffers are allocated; the file is mapped)
    warm   the same call again, twice (the minimum): memory and buffers exist

and reports the arena's own account of the cold read (`grow_ms` spent inside
``bb_arena_alloc``, `prepare_ms` spent on the library's background thread
since ``open()``, `prepare_wait_ms` the read waited for it).

Three children:
    clean            the device as the bench finds it
    dirty            the child first takes nearly all of the device's memory,
                     writes it and frees it: the driver clears what was freed,
                     and the NEXT allocation of the process -- whoever makes it,
                     here the arena's step -- waits for that (the 1.5-4.6 s of
                     round 4's DESIGN section 6; profiles/r05b_grow_probe.log,
                     r05d_cold_read.json)
    dirty_noprepare  the same with BB_ARENA_PREPARE=0: the growth inside read()
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(gib, mode, path):
    import numpy as np
    sys.path.insert(0, ROOT)
    t_start = time.perf_counter()
    import torch
    import baseband_amd as bb
    from baseband_amd import arena, synth
    t_import = time.perf_counter() - t_start
    frame, spf = 8032, 32000
    nframes = int(gib * 2 ** 30) // frame
    # the file: cfg2 headers (frame k at second k // 1000, frame_nr k % 1000) + random payloads
    small, h0 = synth.random_vdif(7, 2000, payload_nbytes=8000, frame_rate=1000)
    rng = np.random.default_rng(11)
    with open(path, 'wb') as f:
        step = 20000
        for lo in range(0, nframes, step):
            n = min(step, nframes - lo)
            img = rng.integers(0, 256, size=(n, frame), dtype=np.uint8)
            words = img.view(np.uint32).reshape(n, frame // 4)
            k = np.arange(lo, lo + n, dtype=np.uint64)
            w = [int(x) for x in h0.words]
            words[:, 0] = (w[0] + k // 1000).astype(np.uint32)
            words[:, 1] = ((w[1] & 0xff000000) + k % 1000).astype(np.uint32)
            words[:, 2] = w[2]
            words[:, 3] = w[3]
            words[:, 4:8] = 0
            f.write(img.tobytes())
    size = os.path.getsize(path)
    with open(path, 'rb') as f:                      # page cache warm
        while f.read(64 << 20):
            pass
    t0 = time.perf_counter()
    torch.zeros(1, device='cuda')
    torch.cuda.synchronize()
    t_ctx = time.perf_counter() - t0
    # the library's kernels and tables: a small read from memory (code objects load at first launch)
    import io
    t0 = time.perf_counter()
    with bb.vdif.open(io.BytesIO(small.tobytes()), 'rs', sample_rate=32e6) as fh:
        fh.read()
    torch.cuda.synchronize()
    t_small = time.perf_counter() - t0

    dirtied = None
    if mode.startswith('dirty'):
        free_b, total_b = torch.cuda.mem_get_info()
        n = int(free_b - (12 << 30))
        t0 = time.perf_counter()
        x = torch.empty(n, dtype=torch.uint8, device='cuda')
        x.fill_(7)
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
        dirtied = {"GiB": round(n / 2 ** 30, 1), "alloc_fill_free_ms": round((time.perf_counter() - t0) * 1e3, 1)}

    def one():
        t = time.perf_counter()
        fh = bb.vdif.open(path, 'rs', sample_rate=32e6)
        t_open = time.perf_counter()
        got = fh.read()
        torch.cuda.synchronize()
        t_read = time.perf_counter()
        fh.close()
        return got, (t_open - t) * 1e3, (t_read - t_open) * 1e3

    got, open_ms, read_ms = one()
    ar = arena.default()
    st = ar.stats() if ar is not None else {}
    in_arena = bool(ar is not None and ar.owns(got))
    ok = bool(got.numel() == nframes * spf)
    # spot check: the last frame against the level table
    from baseband_amd import _lib
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    with open(path, 'rb') as f:
        f.seek((nframes - 1) * frame + 32)
        raw = np.frombuffer(f.read(8000), np.uint8)
    exp = lev[(raw[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3].reshape(-1)
    ok &= bool(np.array_equal(got[-spf:].cpu().numpy().view(np.uint32), exp.view(np.uint32)))
    warm = []
    for _ in range(2):
        del got
        got, o_ms, r_ms = one()
        warm.append((o_ms, r_ms))
    w_open, w_read = min(warm, key=lambda p: p[0] + p[1])
    out = {"mode": mode, "file_GiB": round(size / 2 ** 30, 3), "output_GB": round(nframes * spf * 4 / 1e9, 2),
           "import_s": round(t_import, 2), "hip_context_ms": round(t_ctx * 1e3, 1),
           "first_small_read_ms": round(t_small * 1e3, 1), "dirtied": dirtied,
           "cold": {"open_ms": round(open_ms, 2), "read_ms": round(read_ms, 2)},
           "warm": {"open_ms": round(w_open, 2), "read_ms": round(w_read, 2)},
           "cold_minus_warm_ms": round(open_ms + read_ms - w_open - w_read, 2),
           "output_in_arena": in_arena, "spot_check": ok,
           "arena": {k: (round(st[k], 2) if isinstance(st.get(k), float) else st.get(k))
                     for k in ("grow_ms", "prepares", "prepare_ms", "prepare_wait_ms", "bytes_backed", "steps",
                               "last_probe_gbps")},
           "prepare": os.environ.get('BB_ARENA_PREPARE', '1') not in ('0', 'off', 'no')}
    print("COLD_READ_JSON " + json.dumps(out), flush=True)


def leg_cold_first_read(gib=2.0, timeout=240):
    """Run the children (the caller has NOT touched the GPU yet).  Returns the
    leg's dict; a child that fails is reported in its slot."""
    tmp_root = os.environ.get('TMPDIR', '/tmp')
    try:
        import shutil
        free_b = shutil.disk_usage(tmp_root).free
    except OSError:
        free_b = 0
    if free_b < int(gib * 2 ** 30) + (1 << 30):
        # (an environment matter, not a result: the leg is skipped and counts for no check)
        return {"skipped": "{} has {:.1f} GiB free, a {:.1f} GiB temporary file does not fit".format(
            tmp_root, free_b / 2 ** 30, gib)}
    res = {"what": "fresh child process: first open(2 GiB cfg2 file).read() against the same call again "
                   "(wall ms incl. sync; page cache warm; the library's kernels loaded by a small read before)",
           "runs": []}
    for mode, env_extra in (("clean", {}), ("dirty", {}), ("dirty_noprepare", {"BB_ARENA_PREPARE": "0"})):
        fd, path = tempfile.mkstemp(prefix='bb_cold_', suffix='.vdif', dir=tmp_root)
        os.close(fd)
        env = dict(os.environ, **env_extra)
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', repr(gib), mode, path],
                               env=env, capture_output=True, text=True, timeout=timeout)
            row = None
            for ln in r.stdout.splitlines():
                if ln.startswith('COLD_READ_JSON '):
                    row = json.loads(ln[len('COLD_READ_JSON '):])
            if row is None:
                row = {"mode": mode, "error": "rc {}: {}".format(r.returncode, (r.stderr or r.stdout)[-400:])}
        except Exception as exc:
            row = {"mode": mode, "error": repr(exc)[:300]}
        finally:
            try:
                os.remove(path)
            except OSError:
                pass
        res["runs"].append(row)
    by = {r.get("mode"): r for r in res["runs"]}
    if "cold_minus_warm_ms" in by.get("clean", {}):
        res["cold_minus_warm_ms"] = by["clean"]["cold_minus_warm_ms"]
    if "cold_minus_warm_ms" in by.get("dirty", {}):
        res["cold_minus_warm_ms_dirty_memory"] = by["dirty"]["cold_minus_warm_ms"]
    if "cold_minus_warm_ms" in by.get("dirty_noprepare", {}):
        res["cold_minus_warm_ms_dirty_memory_no_prepare"] = by["dirty_noprepare"]["cold_minus_warm_ms"]
    res["all_ok"] = all(r.get("spot_check") is True for r in res["runs"])
    return res


if __name__ == '__main__':
    if len(sys.argv) >= 5 and sys.argv[1] == '--child':
        _child(float(sys.argv[2]), sys.argv[3], sys.argv[4])
    else:
        print(json.dumps(leg_cold_first_read(), indent=1))
