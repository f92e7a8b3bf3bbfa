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
    if dist.get_rank(group) == src_rank:
        buf = src.to(torch.int64).contiguous()
        assert buf.numel() == nentries
    else:
        buf = torch.empty(nentries, dtype=torch.int64,
                          device=device if device is not None else 'cpu')
    dist.broadcast(buf, src=src_rank, group=group)
    return buf


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
