"""bench.py's locate leg alone (8 GiB cfg2 image):  python tools/run_locate_leg.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from baseband_amd import kernels
kernels.init()
dev = torch.device('cuda', 0)
nframes = (8 << 30) // 8032
image, _ = bench.image_buffer(nframes * 8032, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
from baseband_amd import _lib
for blocks in [int(b) for b in os.environ.get("BB_LOCATE_GRIDS", "65536,131072,262144,524288,65536,131072").split(",")]:
    kernels.tune(_lib.TUNE_BLOCKS, blocks)
    r = bench.leg_locate(image, h0, nframes)
    print(json.dumps({"grid_cap": blocks or 16384, "ms": r["ms"], "ms_median": r["ms_median"], "TBps": round(r["algorithmic_GBps"] / 1e3, 3), "ok": r["spot_check"]}))
kernels.tune(_lib.TUNE_BLOCKS, 0)
