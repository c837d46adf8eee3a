"""Times jmhip_deblock_frame (strength + wavefront kernels) with the context's own HIP events.
Usage: python tools/time_deblock.py [w h [worst|typical|idle [fmt [nslices]]]]
  worst: every edge has bS > 0 (random cbp, 15 % intra);  typical: P-picture statistics (5 % intra, 20 % of the blocks coded, smooth motion);
  idle: no edge needs filtering (bS 0 everywhere): the cost of the walk itself."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge._load_pkg()
from tests.test_deblock import make_case

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1088)
kind = sys.argv[3] if len(sys.argv) > 3 else "worst"
fmt = int(sys.argv[4]) if len(sys.argv) > 4 else 1
nslices = int(sys.argv[5]) if len(sys.argv) > 5 else 1        # > 1: that many row slices with disable_idc 2 (independent bands)          # 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4, 0 = 4:0:0
rng = np.random.default_rng(0)
planes, mbs, blks = make_case(pkg, rng, w, h, fmt, idc_mode="two" if nslices > 1 else "zero", nslices=max(nslices, 1), intra_frac={"worst": 0.15, "typical": 0.05, "idle": 0.0}[kind])
if kind != "worst":
    blks["mv"][:] = (3, -2)
    blks["ref_id"][:, 0] = 0
    keep = rng.random(mbs.size) < (0.2 if kind == "typical" else 0.0)
    mbs["cbp_blk"] = np.where(keep, mbs["cbp_blk"], 0)
ctx = pkg.Context(w, h, yuv_format=fmt, max_refs=1, search_range=8)
ctx.recon_upload(*planes)
ctx.deblock_frame(mbs, blks)
ctx.timing_enable(True)
ctx.timing_read()
for _ in range(10):
    ctx.deblock_frame(mbs, blks)
ctx.sync()
ms, n = ctx.timing_read()["deblock"]
print("%s fmt%d slices%d %dx%d: deblock %.3f ms per picture (%d launches), %d diagonals" % (kind, fmt, nslices, w, h, ms / n, n, w // 16 + 2 * (h // 16 - 1)))
ctx.close()
