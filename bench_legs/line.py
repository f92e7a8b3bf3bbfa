"""The self-checks of a run and the line bench.py prints."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

FORCE_FAIL_ENV = 'BB_BENCH_FORCE_CHECK_FALSE'       # tests: make the named check read false
CHECKS_RC = 3                                       # exit status when a check of the line is false


def collect_checks(line):
    """Every self-check the line carries, folded into one verdict.  Returns
    (checks_ok, {name: bool}).  A leg that is in the line but failed before
    its check was evaluated counts as false: a check that did not run is not a
    passed check.  ``BB_BENCH_FORCE_CHECK_FALSE=<name>`` forces one false
    (tests/test_bench_cli.py: the exit status must follow)."""
    checks = {}

    def put(name, leg, *path):
        v = leg
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        checks[name] = v is True

    put("headline.sanity_spot_check", line, "sanity_spot_check")
    if "parity_digests" in line:
        put("parity_digests.all_match", line, "parity_digests", "all_match")
    if "invalid_fill" in line:
        put("invalid_fill.flagged_frame_is_fill", line, "invalid_fill", "flagged_frame_is_fill")
        put("invalid_fill.neighbour_frame_is_data", line, "invalid_fill", "neighbour_frame_is_data")
        put("invalid_fill.all_invalid_output_is_fill", line, "invalid_fill", "all_frames_invalid", "output_is_fill")
    if "cfg3" in line:
        if line.get("dry_run"):
            put("cfg3.index_ok", line, "cfg3", "index_ok")
        else:
            put("cfg3.sanity_spot_check", line, "cfg3", "sanity_spot_check")
            if isinstance(line["cfg3"], dict) and "rank_local_scan" in line["cfg3"]:
                put("cfg3.rank_local_scan.index_equals_broadcast", line, "cfg3", "rank_local_scan",
                    "index_equals_broadcast")
    if "pipeline" in line and not (isinstance(line["pipeline"], dict) and "skipped" in line["pipeline"]):
        put("pipeline.all_match", line, "pipeline", "all_match")
        sw = _get(line, "pipeline", "sequence_writer")
        if isinstance(sw, dict):
            put("pipeline.sequence_writer.read_back_matches", line, "pipeline", "sequence_writer", "read_back_matches")
    if "cold_first_read" in line and not (isinstance(line["cold_first_read"], dict) and "skipped" in line["cold_first_read"]):
        put("cold_first_read.all_ok", line, "cold_first_read", "all_ok")
    ap = _get(line, "roofline", "arena_placed_output")
    if isinstance(ap, dict) and "skipped" not in ap:
        put("roofline.arena_placed_output.spot_check", line, "roofline", "arena_placed_output", "spot_check")
    if "other_configs" in line:
        oc = line["other_configs"]
        checks["other_configs.spot_checks"] = bool(oc) and all(
            isinstance(c, dict) and "error" not in c and c.get("spot_check", True) is True for c in oc)
    forced = os.environ.get(FORCE_FAIL_ENV)
    if forced:
        checks[forced] = False
    return all(checks.values()), checks


LINE_LIMIT = 2000                                   # bytes; the driver keeps the tail of stdout only
DETAIL_NAME = 'bench_detail.json'

# short names of the `other_configs` rows in the line's `secondary` block
_SECONDARY_KEYS = (
    ("bb_encode_flat VDIF 2", "enc2"), ("bb_encode_flat VDIF 4", "enc4"), ("bb_encode_flat VDIF 8", "enc8"),
    ("sample.vdif layout", "vdif_8thr"), ("Mark 5B", "mark5b"), ("Mark 4", "mark4"),
    ("channels first", "guppi_cf"), ("time first, channel list", "guppi_tf_pick"), ("time first", "guppi_tf"),
    ("flat int8", "dada_i8"), ("MKBF", "mkbf"), ("NBIT=32", "dada32_copy"),
    ("VDIF 1-bit", "vdif_1bit"), ("VDIF 4-bit", "vdif_4bit"), ("VDIF 8-bit", "vdif_8bit"),
    ("GSB", "gsb_4bit"), ("subset of 2 of 16", "gather_select"),
    ("bb_vdif_locate", "locate"))


def _get(d, *path):
    for k in path:
        d = d.get(k) if isinstance(d, dict) else None
    return d


def secondary_summary(line):
    """{short name: fraction of 8 TB/s (kernel legs) or GB/s of file bytes
    (pipeline)} of the legs that ran; the full rows are in the detail file."""
    sec = {}
    v = _get(line, "cfg3", "roofline", "frac")
    if v is not None:
        sec["cfg3"] = v
    v = _get(line, "api_read", "ms_over_kernel_leg")
    if v is not None:
        sec["api_read_over_kernel"] = v
    for c in line.get("other_configs") or []:
        if isinstance(c, dict) and "frac" in c:
            for pat, key in _SECONDARY_KEYS:
                if pat in c.get("case", ""):
                    sec[key] = c["frac"]
                    break
    pl = _get(line, "pipeline", "formats")
    if pl:
        r = [f.get("file_GBps_best") for f in pl if isinstance(f, dict) and f.get("file_GBps_best")]
        w = [f.get("writer_GBps") for f in pl if isinstance(f, dict) and f.get("writer_GBps")]
        if r:
            sec["pipeline_GBps"] = [min(r), max(r)]
        if w:
            sec["writer_GBps"] = [min(w), max(w)]
        sec["pinned_h2d_GBps"] = _get(line, "pipeline", "pinned_h2d_GBps")
    v = _get(line, "pipeline", "sequence_writer", "writer_GBps")
    if v is not None:
        sec["writer_sequence_GBps"] = v
    ms = _get(line, "mid_size", "sizes")
    if ms and isinstance(ms[0], dict):
        sec["mid_2p15_arena_min"] = _get(ms[0], "arena", "frac_min")
        sec["mid_2p15_torch_min"] = _get(ms[0], "torch_empty", "frac_min")
        sec["mid_2p15_api_read"] = _get(ms[0], "api_read", "frac")
        # what explains a low mid-size read: where the arena's step landed (its probe, and
        # whether a second candidate was tried / kept) and what stands around the kernel
        sec["mid_2p15_host_ms"] = _get(ms[0], "api_read", "host_and_scan_ms")
        sec["mid_2p15_api_kernel"] = _get(ms[0], "api_read", "kernel_frac_same_output")
        st = _get(line, "mid_size", "arena_after") or {}
        if st:
            sec["arena_probe_gbps"] = round(st.get("last_probe_gbps") or 0.0, 0)         # the step kept last
            sec["arena_second_chances"] = [st.get("second_chances"), st.get("second_chance_wins"),
                                           st.get("second_chances_no_room")]     # tried, won, wanted but no room
            sec["arena_create_ms"] = round(st.get("last_create_ms") or 0.0, 1)
            if st.get("probe_history"):
                sec["arena_probes"] = st["probe_history"][-6:]          # every candidate probed (kept or not), GB/s
    for k in ("cold_minus_warm_ms", "cold_minus_warm_ms_dirty_memory"):
        v = _get(line, "cold_first_read", k)
        if v is not None:
            sec[k] = v
    def r3(v):
        if isinstance(v, float):
            return round(v, 3)
        if isinstance(v, list):
            return [r3(x) for x in v]
        return v
    return {k: r3(v) for k, v in sec.items() if v is not None}


def compact_line(line, detail=DETAIL_NAME):
    """THE line: the contract's keys, `roofline`, `cpu_baseline`, `checks_ok`
    and a short `secondary` block, at most LINE_LIMIT bytes (VERDICT r4 next 1:
    the driver keeps only the tail of stdout, a 21 KB line was unreadable to
    it).  Everything else goes to the detail file."""
    c = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                  "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = line.get("config") or {}
    c["config"] = {k: cfg[k] for k in ("workload", "input_memory", "output_memory") if k in cfg}
    rf = line.get("roofline") or {}
    c["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms_avg",
                                            "algorithmic_bytes_per_launch", "traffic")}
    c["roofline"]["traffic_over_algorithmic"] = _get(rf, "traffic_detail", "traffic_over_algorithmic")
    c["roofline"]["kernel_time_over_write_only_time"] = _get(rf, "measured_write_only",
                                                             "kernel_time_over_write_only_time")
    v = _get(rf, "arena_placed_output", "frac")
    if v is not None:
        c["roofline"]["frac_arena_placed_output"] = v
    cpu = line.get("cpu_baseline")
    if isinstance(cpu, dict):
        c["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}
        c["cpu_baseline"]["sample"] = str(cpu.get("sample_short") or cpu.get("sample") or "")[:72]
        c["cpu_baseline"]["all_cores"] = {"value": _get(cpu, "all_cores", "value"),
                                          "cores": _get(cpu, "all_cores", "cores")}
        c["cpu_baseline"]["reference_as_written_estimate"] = _get(cpu, "calibration",
                                                                  "reference_as_written_estimate_Msps")
    else:
        c["cpu_baseline"] = cpu
    for k in ("dry_run", "ranks_seen", "per_rank"):
        if k in line:
            c[k] = line[k]
    c3 = line.get("cfg3")
    if isinstance(c3, dict) and (line.get("n_gpus") or 1) > 1:
        # what an N > 1 run must show of the sharded leg: every rank took part in the ONE
        # collective, and what rank 0 (which holds the whole file) had to fit
        c["cfg3"] = {"collective": {k: _get(c3, "collective", k) for k in ("ranks_seen", "bytes", "ms", "backend")},
                     "rank0_footprint_GiB": c3.get("rank0_footprint_GiB")}
        for k in ("value", "ms_per_step"):
            if c3.get(k) is not None:
                c["cfg3"][k] = c3[k]
    c["checks_ok"] = line.get("checks_ok")
    failed = [k for k, v in (line.get("checks") or {}).items() if not v]
    if failed:
        c["failed_checks"] = failed
    sec = secondary_summary(line)
    if sec:
        c["secondary"] = sec
    c["detail"] = detail
    # never above the limit: shed the optional blocks, largest first, then cut
    # the strings shorter and shorter
    def size():
        return len(json.dumps(c, separators=(',', ':')).encode())

    def clip(x, n):
        if isinstance(x, dict):
            return {k: clip(v, n) for k, v in x.items()}
        return x[:n] if isinstance(x, str) else x

    # (secondary figures go one at a time, the rows that repeat a sibling first: the block as a
    # whole is what explains a low headline or mid-size read and is shed last)
    for k in ("vdif_4bit", "vdif_1bit", "gsb_4bit", "mkbf", "guppi_tf", "enc8", "dada_i8", "vdif_8thr", "mark4",
              "writer_sequence_GBps", "pinned_h2d_GBps", "arena_create_ms", "arena_first_probe_gbps", "arena_probe_gbps", "enc2",
              "guppi_tf_pick", "locate", "mid_2p15_api_kernel"):
        if size() <= LINE_LIMIT:
            break
        c.get("secondary", {}).pop(k, None)
    for drop in ("secondary", "per_rank", "failed_checks"):
        if size() <= LINE_LIMIT:
            break
        c.pop(drop, None)
    for n in (160, 80, 40, 16):
        if size() <= LINE_LIMIT:
            break
        for k in list(c):
            c[k] = clip(c[k], n)
    return c


def write_detail(line, path=None):
    """The full record of the run next to the script (tools/prof_round.sh copies
    it into profiles/); /tmp when the tree is read-only.  Returns the path
    written or None."""
    for p in ([path] if path else []) + [os.path.join(ROOT, DETAIL_NAME), os.path.join('/tmp', DETAIL_NAME)]:
        try:
            with open(p, 'w') as f:
                json.dump(line, f, indent=1)
                f.write('\n')
            return p
        except OSError:
            continue
    return None


def finish(line, detail_path=None):
    """Attach `checks_ok` / `checks`, write the detail file, print THE line
    (compact, last thing on stdout), return the exit status."""
    ok, checks = collect_checks(line)
    line["checks_ok"] = ok
    line["checks"] = checks
    where = write_detail(line, detail_path)
    # (relative to the repository when it lies inside: the line has 2,000 bytes)
    shown = where or "not written"
    if where and os.path.abspath(where).startswith(ROOT + os.sep):
        shown = os.path.relpath(where, ROOT)
    c = compact_line(line, detail=shown)
    sys.stderr.flush()
    print(json.dumps(c, separators=(',', ':')), flush=True)
    if not ok:
        print("bench.py: self-checks FAILED: " + ", ".join(k for k, v in checks.items() if not v),
              file=sys.stderr, flush=True)
        return CHECKS_RC
    return 0
