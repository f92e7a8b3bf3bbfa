#!/usr/bin/env python3
"""Headline benchmark: decoded Msamples/s + achieved HBM GB/s, VDIF 2-bit.

One step = one pass of the hot path (header scan -> index build -> packed
sample decode) over one synthetic file image that is already resident in HBM:
BASELINE.json configs[1], "synthetic 8 GiB single-thread VDIF, 2-bit real,
1 channel, EDV 0" (8032-byte frames, 32000 samples per frame).

``python bench.py --gpus N``: one process per GPU.  Under
``python -m torch.distributed.run`` (the driver's way) the ranks come from the
environment; started plainly with N > 1 this process -- before it touches the
GPU -- starts ``torch.distributed.run`` with N workers itself and passes their
output through.  Every rank decodes its own time slab of the same size (weak
scaling, no data-path collective: frames are independent, SURVEY.md section
8e); rank 0 prints ONE JSON line.

Besides the contract's keys the line carries
  roofline      dominant kernel (named by the library: bb_last_kernel) timed
                with HIP events on the launching stream; `traffic` = HBM bytes
                per launch from two rocprofv3 --pmc child passes of this same
                script (run before this process touches the GPU), or the
                committed profiles/traffic_latest.json when that fails
  cpu_baseline  the NumPy restatement of the reference's per-frame loop
                (oracle/, kind "port") on a bounded sample, rank 0 at N = 1,
                with the calibration against the real reference
                (tests/golden/cpu_calibration.json)
  api_read      the same 8 GiB image through the drop-in API:
                ``vdif.open(<device tensor>, 'rs').read(out=out)``
  invalid_fill  the headline step with `invalid_data` set in 1 % of the frames
                (the fill path must cost nothing)
  cfg3          BASELINE configs[2] (8-thread 2-bit complex 16-channel VDIF):
                rank 0 scans the whole file and builds the frame index, ONE
                broadcast (RCCL) replicates it, every rank decodes its slab
  parity_digests sha256 of golden files decoded here vs the reference's digests
  other_configs Mark 5B / Mark 4 / GUPPI / DADA / 8-thread real VDIF kernels
                on inputs that decode to the headline's output size (N = 1 only)
  pipeline      file -> HBM -> decode: open(path).read() of 2 GiB files (VDIF cfg2 /
                cfg3, Mark 5B, Mark 4, GUPPI, DADA) from the page cache, GB/s of
                file bytes against the pinned H2D rate, per-window times
  mid_size      the launch sizes ordinary read() calls issue (2^15, 2^16, 2^18
                cfg2 frames; GUPPI 8 GiB in) into FRESH outputs, torch.empty
                against the placement arena the readers allocate from, min /
                median / max over the draws (N = 1 only)
  cold_first_read fresh child processes: the first open(2 GiB file).read() of a
                process against the same call again; on a clean device, and
                after the child dirtied nearly all of the device's memory
                (with and without the arena's background growth)

The LAST stdout line is a compact JSON object (at most 2,000 bytes: the
contract's keys, `roofline`, `cpu_baseline`, `checks_ok`, a short `secondary`
block); the full record of every leg goes to ``bench_detail.json`` next to this
script.  The legs live in ``bench_legs/``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (names other scripts and the tests take from this module)
from bench_legs.common import (HBM_PEAK_GBS, FRAME_NBYTES, HEADER_NBYTES, PAYLOAD_NBYTES, SPF, FRAME_RATE,   # noqa: E402,F401
                               CFG3_THREADS, CFG3_NCHAN, CFG3_ORDER, CFG3_SET_RATE, _s32,
                               make_file_image_on_device, empty_with_patience, image_buffer,
                               timed_launches, expand_2bit, _free_port, _git_commit, _run_group)
from bench_legs.cpu import cpu_baseline, physical_cores                          # noqa: E402,F401
from bench_legs.traffic import live_traffic, file_traffic                        # noqa: E402,F401
from bench_legs.line import (collect_checks, compact_line, finish, FORCE_FAIL_ENV, CHECKS_RC,    # noqa: E402,F401
                             LINE_LIMIT, DETAIL_NAME)
from bench_legs.ranks import spawn_ranks, dry_run, CPU_JSON_ENV                  # noqa: E402,F401
from bench_legs.api import parity_digests, leg_api_read, leg_invalid_fill        # noqa: E402,F401
from bench_legs.cfg3 import leg_cfg3                                             # noqa: E402,F401
from bench_legs.other_configs import leg_locate, leg_other_configs               # noqa: E402,F401
from bench_legs.pipeline import pinned_h2d_rate, leg_pipeline                    # noqa: E402,F401
from bench_legs.mid_size import leg_mid_size                                     # noqa: E402,F401
from bench_legs.cold_read import leg_cold_first_read                             # noqa: E402,F401
from bench_legs.arena_headline import leg_arena_headline                         # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--gib', type=float, default=8.0, help="file image size per GPU")
    ap.add_argument('--cfg3-gib', type=float, default=8.0, help="cfg3 leg: file bytes per GPU")
    ap.add_argument('--pipeline-gib', type=float, default=2.0, help="pipeline leg: size of each temporary file")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--traffic', choices=('live', 'file', 'none'), default='live')
    ap.add_argument('--no-extra-legs', action='store_true',
                    help="skip api_read / cfg3 / other_configs (headline only); same as --legs headline")
    ap.add_argument('--legs', choices=('all', 'headline', 'cfg3'), default='all',
                    help="all (default); headline: the headline step only, no CPU baseline, no counter passes "
                         "(an N-GPU run finishes in well under a minute); cfg3: headline + the cfg3 leg")
    ap.add_argument('--pmc-child', action='store_true',
                    help="internal: headline kernel only, no JSON extras (run under rocprofv3 --pmc)")
    ap.add_argument('--force-dist', action='store_true',
                    help="initialise the RCCL process group even with one rank, so that every "
                         "collective of the multi-rank path runs (a check of the N > 1 code on a 1-GPU box)")
    ap.add_argument('--detail', default=None,
                    help="where the full record goes (default: bench_detail.json next to this script)")
    ap.add_argument('--dry-run', action='store_true',
                    help="CPU rehearsal of the multi-rank plumbing with gloo (no GPU, no measurement)")
    args = ap.parse_args()
    if args.pmc_child:
        args.no_cpu_baseline, args.no_extra_legs, args.traffic = True, True, 'none'
    if args.legs == 'headline':
        args.no_cpu_baseline, args.no_extra_legs, args.traffic = True, True, 'none'
    elif args.legs == 'cfg3':
        args.no_cpu_baseline, args.traffic = True, 'none'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not under a launcher: become one (nothing here has touched the GPU)
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.dry_run:
        return dry_run(args, rank, world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus {} but WORLD_SIZE is {}".format(args.gpus, world))
    # host-side legs run first: they fork / start child processes, which must
    # happen before this process initialises the GPU
    cpu = None
    traffic_detail = None
    # (is there a GPU?  Asked WITHOUT initialising the HIP runtime: the legs
    # below fork workers and start profiler children, and a forked child must
    # not inherit a live runtime -- torch.cuda.device_count() can initialise it
    # on this build)
    have_gpu = os.path.exists('/dev/kfd')
    if rank == 0 and have_gpu and not args.no_cpu_baseline:
        handed = os.environ.get(CPU_JSON_ENV)
        if handed and os.path.exists(handed):
            with open(handed) as f:
                cpu = json.load(f)                  # timed by the parent that started the ranks
        else:
            # N = 1, or N > 1 under an external launcher: rank 0 times it before it
            # touches the GPU (the other ranks wait in the rendezvous meanwhile)
            cpu = cpu_baseline()
            if world > 1:
                cpu["timed_by"] = "rank 0 before initialising the GPU; the other ranks waited in the rendezvous"
    if rank == 0 and world > 1:
        traffic_detail = {"hbm_bytes_per_launch": None,
                          "reason": "counter passes run at N = 1 only: they are child runs of this script "
                                    "under rocprofv3 --pmc on the first GPU, which the ranks of an N > 1 "
                                    "run are about to use; the N = 1 line of the same commit carries the counters"}
    if rank == 0 and world == 1 and have_gpu:
        if abs(args.gib - 8.0) < 1e-9 and args.traffic != 'none':
            if args.traffic == 'live':
                try:
                    traffic_detail = live_traffic(args.gib)
                except Exception as exc:
                    traffic_detail = {"live_error": repr(exc)[:400]}
            if traffic_detail is None or "hbm_bytes_per_launch" not in traffic_detail:
                try:
                    err = (traffic_detail or {}).get("live_error")
                    traffic_detail = file_traffic()
                    if err:
                        traffic_detail["live_error"] = err
                except Exception:
                    pass
    cold = None
    if rank == 0 and world == 1 and have_gpu and args.legs == 'all' and not args.no_extra_legs:
        try:        # fresh child processes on the GPU this process has not touched yet
            cold = leg_cold_first_read()
        except Exception as exc:
            cold = {"error": repr(exc)[:400]}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback).")
    if world > torch.cuda.device_count():
        raise SystemExit("bench.py: {} ranks but only {} GPUs visible".format(world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        import datetime
        # (a rank that fails must become an error on the others, not a hang)
        # (RCCL prints a version banner on the C library's stdout when the communicator is
        # made; the contract is ONE JSON line there: the banner goes to stderr)
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=device, timeout=datetime.timedelta(seconds=240))
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)               # (communicators are made lazily on some builds)
            torch.cuda.synchronize()
        finally:
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)

    from baseband_amd import kernels, _lib
    from baseband_amd.parallel import frame_slab
    kernels.init()

    nframes = int(args.gib * 2 ** 30) // FRAME_NBYTES
    # time-slab sharding: rank r owns frames [r*nframes, (r+1)*nframes)
    first_frame, _ = frame_slab(nframes * world, rank, world)
    image, image_memory = image_buffer(nframes * FRAME_NBYTES, device)
    image, h0 = make_file_image_on_device(nframes, 12345 + rank, first_frame, device, into=image)
    pattern, mask = h0.invariant_pattern()
    out = empty_with_patience(nframes * SPF, torch.float32, device)
    bytes_in = nframes * FRAME_NBYTES
    bytes_out = nframes * SPF * 4
    alg_bytes = bytes_in + bytes_out

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(k=None):
        recs = kernels.vdif_scan(image, nframes, FRAME_NBYTES, HEADER_NBYTES, pattern,
                                 mask, h0['seconds'], h0['frame_nr'] + first_frame,
                                 FRAME_RATE)
        src = kernels.build_index(recs, nframes, 1, None)
        if k is not None:
            ev[k][0].record()
        kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2,
                              src=src, out=out)
        if k is not None:
            ev[k][1].record()
        return recs, src

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kernel_name = _lib.last_kernel()

    # sanity spot check (outside the timed region; NOT the parity proof, which
    # lives in tests/): three frames re-expanded on the host from the library's
    # own level table
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    ok = True
    if args.warmup:
        for f in (0, nframes // 3, nframes - 1):
            raw = image[f * FRAME_NBYTES + HEADER_NBYTES:(f + 1) * FRAME_NBYTES].cpu().numpy()
            got = out[f * SPF:(f + 1) * SPF].cpu().numpy()
            ok &= bool(np.array_equal(got.view(np.uint32), expand_2bit(raw, lev).view(np.uint32)))

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg = sum(kern_ms) / len(kern_ms)
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        allk = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(allk, torch.tensor([kern_avg], dtype=torch.float64, device=device))
        per_rank = [float(x.item()) for x in allk]
    if args.pmc_child:
        return

    achieved = alg_bytes / (kern_avg * 1e-3) / 1e9
    total_samples = nframes * SPF * world * args.steps
    value = total_samples / elapsed / 1e6
    traffic = None
    if traffic_detail and traffic_detail.get("hbm_bytes_per_launch"):
        traffic = traffic_detail["hbm_bytes_per_launch"]
        traffic_detail["traffic_over_algorithmic"] = round(traffic / alg_bytes, 4)
        if traffic < 0.98 * alg_bytes:
            # fewer bytes than the kernel must move: the counter rows are not the
            # headline launches' (say so rather than report an impossible figure)
            traffic_detail["rejected"] = "counters below the algorithmic bytes: {:.4g} B".format(traffic)
            traffic = None

    line = {
        "metric": "decoded Msamples/s, VDIF 2-bit 1-thread (scan + index + decode, input resident in HBM)",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "float32", "data": "synthetic",
        "config": {"workload": "cfg2: synthetic {:.3f} GiB per GPU single-thread VDIF, "
                               "2-bit real, 1 channel, EDV 0, 8032-byte frames"
                               .format(bytes_in / 2 ** 30),
                   "frames_per_gpu": nframes, "bytes_in_per_gpu": bytes_in,
                   "bytes_out_per_gpu": bytes_out,
                   "input": "packed 2-bit codes (uint8 file image; memory: {})".format(image_memory),
                   "input_memory": image_memory,
                   "output": "float32 samples: "
                   "full-size tensor kept in HBM (no slab recycling)",
                   "output_memory": "torch.empty, {:.1f} GiB (above the readers' 64 GiB arena block limit)"
                                    .format(bytes_out / 2 ** 30),
                   "sharding": "time slabs, one per rank, no collective"},
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms_avg": round(kern_avg, 4),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "traffic": traffic, "traffic_detail": traffic_detail},
        "sanity_spot_check": ok,
    }
    if per_rank is not None:
        line["per_rank"] = {"kernel_ms_avg": [round(x, 4) for x in per_rank],
                            "roofline_frac": [round(alg_bytes / (x * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                              for x in per_rank]}
    if not args.no_extra_legs:
        # (each leg is fenced: a failure is reported in its slot, the headline stands)
        if rank == 0 and world > 1:
            # the legs below include a collective; should a rank die inside it
            # the watchdog takes this process down before the line is printed,
            # so leave the measured headline in the log first (stderr: the
            # contract's ONE line on stdout is printed at the end)
            print("bench.py: headline before the extra legs: " + json.dumps(line),
                  file=sys.stderr, flush=True)
        if rank == 0 and args.legs == 'all':
            try:
                line["parity_digests"] = parity_digests()
            except Exception as exc:
                line["parity_digests"] = {"error": repr(exc)[:300]}
            try:
                line["api_read"] = leg_api_read(args, image, out, kern_avg)
            except Exception as exc:
                line["api_read"] = {"error": repr(exc)[:500]}
            try:
                line["invalid_fill"] = leg_invalid_fill(args, image, out, nframes, h0, first_frame, kern_avg)
                # the fraction of the spec peak AND of what the device
                # was measured to do -- this launch's stores without its reads
                w = line["invalid_fill"]["all_frames_invalid"]
                line["roofline"]["measured_write_only"] = {
                    "GBps": w["write_GBps"], "frac_of_peak": w["frac_of_peak"],
                    "kernel_time_over_write_only_time": w["headline_kernel_ms_over_this"],
                    "what": "the headline launch with every index entry -1 (fill): same kernel, same stores, no reads"}
            except Exception as exc:
                line["invalid_fill"] = {"error": repr(exc)[:500]}
        locate = None
        if rank == 0 and world == 1 and args.legs == 'all':
            try:
                locate = leg_locate(image, h0, nframes)
            except Exception as exc:
                locate = {"case": "bb_vdif_locate", "error": repr(exc)[:500]}
        del image
        try:
            line["cfg3"] = leg_cfg3(args, rank, world, device, dist, out)
        except Exception as exc:
            # (with several ranks the others run into the collective's timeout
            # and land here too; the headline measured above still gets printed)
            line["cfg3"] = {"error": repr(exc)[:500]}
        if rank == 0 and world == 1 and args.legs == 'all':
            try:
                line["other_configs"] = []
                leg_other_configs(device, out, res=line["other_configs"])
            except Exception as exc:
                line["other_configs"].append({"error": repr(exc)[:500]})
            if locate is not None:
                line["other_configs"].append(locate)
            # the launch sizes of ordinary read() calls, into fresh outputs: the
            # 127.5 GiB headline output goes first, the 8 GiB image comes back
            del out
            torch.cuda.empty_cache()
            try:
                image, _ = image_buffer(nframes * FRAME_NBYTES, device)
                image, _ = make_file_image_on_device(nframes, 12345, 0, device, into=image)
                line["mid_size"] = leg_mid_size(device, image)
                del image
            except Exception as exc:
                line["mid_size"] = {"error": repr(exc)[:500]}
            torch.cuda.empty_cache()
            try:
                line["pipeline"] = leg_pipeline(device, gib=args.pipeline_gib)
            except Exception as exc:
                line["pipeline"] = {"error": repr(exc)[:500]}
            torch.cuda.empty_cache()
            try:        # (last: the 127.5 GiB plain output is gone, an arena block of that size fits)
                line["roofline"]["arena_placed_output"] = leg_arena_headline(device, gib=args.gib)
            except Exception as exc:
                line["roofline"]["arena_placed_output"] = {"error": repr(exc)[:500]}
    rc = 0
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if cold is not None:
            line["cold_first_read"] = cold
        rc = finish(line, args.detail)
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == '__main__':
    sys.exit(main() or 0)

