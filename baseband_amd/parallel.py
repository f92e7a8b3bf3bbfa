"""Frame-parallel sharding across the GPUs of one node (one process per GPU).

Frames (frame sets) are independent units: rank r decodes a contiguous time
slab of frame sets into its own HBM and the logical output is the list of the
per-rank tensors -- the 16x expanded samples never cross xGMI (SURVEY.md
section 8e).  No data-path collective is needed for fixed-stride
single-thread files: every rank computes its slab locally.

For multi-thread VDIF the on-disk thread order is arbitrary and frames may be
missing, so the dense frame index built by the scanning rank
(bb_vdif_scan + bb_build_index) is replicated with ONE broadcast (RCCL over
xGMI when the backend is "nccl"; the same code runs on "gloo" for CPU tests):
``nsets * nslot`` int64 payload offsets, about 8.5 MB per 8 GiB of 8032-byte
frames.
"""
import torch


def frame_slab(nsets_total, rank, world):
    """Contiguous slab [lo, hi) of frame sets owned by `rank`; sizes differ
    by at most one."""
    base, extra = divmod(int(nsets_total), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def broadcast_frame_index(src, nentries, src_rank=0, device=None, group=None):
    """Replicate the dense source table (int64, `nentries` long) from
    `src_rank` to every rank.  Non-source ranks pass ``src=None``."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return src
    # "nccl" (RCCL) moves device tensors; "gloo" -- CPU tests, and ranks that share
    # ONE GPU, which RCCL refuses (tests/test_multiprocess_gpu.py) -- goes through
    # host memory: 8 bytes per frame, once per file
    via_host = dist.get_backend(group) == 'gloo'
    if dist.get_rank(group) == src_rank:
        buf = src.to(torch.int64).contiguous()
        assert buf.numel() == nentries
        if device is None:
            device = buf.device
        if via_host:
            buf = buf.cpu()
    else:
        buf = torch.empty(nentries, dtype=torch.int64,
                          device='cpu' if via_host or device is None else device)
    dist.broadcast(buf, src=src_rank, group=group)
    return buf if device is None else buf.to(device)


def local_index(src, lo_set, hi_set, nslot, payload_nbytes, align=8):
    """Cut the global index down to the slab [lo_set, hi_set) and rebase the
    payload offsets to the byte range the rank has to stage.

    Returns (local_src, byte_lo, byte_hi): ``local_src`` addresses a buffer
    holding file bytes [byte_lo, byte_hi); missing/invalid entries stay -1.
    """
    part = src[lo_set * nslot:hi_set * nslot]
    valid = part >= 0
    if not bool(valid.any()):
        return part.clone(), 0, 0
    vals = part[valid]
    byte_lo = int(vals.min().item())
    byte_lo -= byte_lo % align
    byte_hi = int(vals.max().item()) + int(payload_nbytes)
    local = torch.where(valid, part - byte_lo, part)
    return local, byte_lo, byte_hi


def sharded_vdif_read(reader, rank=None, world=None, src=None, group=None):
    """Decode this rank's time slab of a (multi-thread) VDIF stream.

    The scanning rank (0) builds the dense frame index with the GPU header
    scan and broadcasts it (ONE collective); every rank then stages only the
    file bytes its slab needs and decodes them into its own HBM.  Returns
    ``(data, (first_sample, last_sample))`` with ``data`` of shape
    ``(nsample_slab, nthread, nchan)`` before squeeze/subset of channels.
    `rank`/`world`/`src` can be given explicitly to run without
    torch.distributed (tests, single process)."""
    import torch.distributed as dist
    use_dist = rank is None and dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank(group) if use_dist else 0
        world = dist.get_world_size(group) if use_dist else 1
    spf = reader.samples_per_frame
    nsets = reader.shape[0] // spf
    nslot = len(reader._thread_ids)
    if src is None:
        if rank == 0:
            src = reader.build_index()
        if use_dist and world > 1:
            src = broadcast_frame_index(src, nsets * nslot, src_rank=0,
                                        device=torch.device('cuda', torch.cuda.current_device()),
                                        group=group)
    lo, hi = frame_slab(nsets, rank, world)
    local, byte_lo, byte_hi = local_index(src, lo, hi, nslot,
                                          reader.header0.payload_nbytes)
    data = reader.decode_with_index(local, byte_lo, byte_hi, hi - lo)
    return data, (lo * spf, hi * spf)


def sharded_read(reader, rank=None, world=None, group=None):
    """This rank's time slab of ANY stream reader (VDIF, Mark 5B, Mark 4,
    GUPPI, DADA, GSB): frames are independent, so rank r simply seeks to the
    start of its slab of frames and reads it through the normal pipeline into
    its own HBM -- no collective at all.  Returns
    ``(data, (first_sample, last_sample))``; the slabs of all ranks tile the
    stream.  (GUPPI files with OVERLAP: a slab starts with its first frame's
    own leading samples, exactly like a seek + read at that offset in the
    reference.)  Use `sharded_vdif_read` when the scan of a multi-thread VDIF file
    should be done once and its index broadcast instead."""
    import torch.distributed as dist
    if rank is None:
        use_dist = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if use_dist else 0
        world = dist.get_world_size(group) if use_dist else 1
    spf = reader.samples_per_frame
    nsample = reader.shape[0]
    nframes = -(-nsample // spf)
    lo, hi = frame_slab(nframes, rank, world)
    first, last = min(lo * spf, nsample), min(hi * spf, nsample)
    reader.seek(first)
    return reader.read(last - first), (first, last)
