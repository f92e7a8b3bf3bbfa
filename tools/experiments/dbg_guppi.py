import sys, os, json, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import numpy as np, torch
from baseband_amd import kernels, _lib, arena, placement
from test_fullsize_gpu import _random_bytes
dev = torch.device('cuda')
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nfr = 8
def run(overlap):
    image = _random_bytes(nfr * blk, 17, dev)[:nfr * blk]
    keep = T - overlap
    out = kernels.decode_i8_tiled(image, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, keep, src0=0, src_stride=blk)
    kn = _lib.last_kernel()
    torch.cuda.synchronize()
    ar = arena.default()
    o = out.view(nfr, keep, npol, nchan, 2)
    b = image.view(torch.int8).view(nfr, nchan, T, npol, 2)[:, :, :keep]
    bad = []
    detail = []
    for f in range(nfr):
        if not torch.equal(o[f].to(torch.float64).sum(0), b[f].to(torch.float64).sum(1).permute(1, 0, 2)):
            bad.append(f)
            wf = b[f].permute(1, 2, 0, 3).to(torch.float32)
            rows = (o[f] != wf).reshape(keep, -1).any(1)
            nzr = torch.nonzero(rows).reshape(-1)
            wrong = o[f][rows]
            # byte offsets of the wrong rows inside the block
            off0 = (f * keep + int(nzr[0])) * npol * nchan * 2 * 4
            off1 = (f * keep + int(nzr[-1]) + 1) * npol * nchan * 2 * 4
            detail.append({"frame": f, "rows_wrong": int(rows.sum()), "first": int(nzr[0]), "last": int(nzr[-1]),
                           "byte_off_MiB": [round(off0 / 2**20, 2), round(off1 / 2**20, 2)],
                           "zero_fraction_of_wrong_rows": round(float((wrong == 0).float().mean()), 4),
                           "nan": int(torch.isnan(wrong).sum())})
    # where does it differ?
    w0 = b[0].permute(1, 2, 0, 3).to(torch.float32)     # (keep, npol, nchan, 2)
    diff_rows = (o[0] != w0).reshape(keep, -1).any(1)
    nz = torch.nonzero(diff_rows).reshape(-1)
    print(json.dumps({"overlap": overlap, "kernel": kn, "ptr": hex(out.data_ptr()), "owned": ar is not None and ar.owns(out),
                      "stats": None if ar is None else {k: ar.stats()[k] for k in ('bytes_backed', 'blocks', 'steps', 'probes', 'bytes_trimmed')},
                      "frames_bad": bad, "detail": detail, "rows_differ_in_frame0": int(diff_rows.sum().item()),
                      "first_last_bad_row": [int(nz[0]), int(nz[-1])] if nz.numel() else None}), flush=True)
print("free", [x >> 30 for x in torch.cuda.mem_get_info()]); run(0)
gc.collect()
print("after test 0:", {k: arena.default().stats()[k] for k in ('bytes_backed', 'blocks', 'bytes_trimmed')})
run(512)
gc.collect()
run(512)
print("free", [x >> 30 for x in torch.cuda.mem_get_info()]); run(0)
