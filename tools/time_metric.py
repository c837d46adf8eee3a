"""Times jmhip_me_frame's general metric path (me_metric.hip) on a 1080p picture:  python tools/time_metric.py [search_range]
One predictor per macroblock (bench.py's recipe), FullSearch, every macroblock and partition; prints ms per frame per metric setting."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge._load_pkg()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H = 1920, 1088
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:H + 32, 0:W + 32]
base = ((np.sin(xx / 6.0) * np.cos(yy / 9.0)) * 70 + 128 + rng.normal(0, 12, (H + 32, W + 32))).clip(0, 255)
ref = base[16:16 + H, 16:16 + W].astype(np.uint8)
cur = base[13:13 + H, 21:21 + W].astype(np.uint8)
cu = [c[::2, ::2].copy() for c in (ref, cur)]
ctx = pkg.Context(W, H, yuv_format=1, max_refs=1, search_range=R)
ctx.ref_upload(0, ref, cu[0], cu[0])
ctx.interp_luma(0)
ctx.interp_chroma(0)
ctx.cur_upload(cur, cu[1], cu[1])
from h264_amd.jmhip import ME_MB_DTYPE
n = (W // 16) * (H // 16)
mbs = np.zeros(n, dtype=ME_MB_DTYPE)
mbs["mb_x"], mbs["mb_y"] = np.arange(n) % (W // 16), np.arange(n) // (W // 16)
mbs["ref_is_0"] = 1
mbs["pred_mv"][:] = rng.integers(-8, 9, (n, 1, 2))
for metric, cme in (((0, 2, 2), 0), ((0, 2, 2), 2), ((0, 0, 0), 0), ((1, 1, 1), 0), ((1, 1, 1), 2), ((2, 2, 2), 0)):
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = 4000
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    prm.metric_set = 1
    prm.metric[0], prm.metric[1], prm.metric[2] = metric
    prm.chroma_me, prm.chroma_me_weight = cme, 1
    ctx.me_frame_async(prm, mbs)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.me_frame_async(prm, None, n)
    ctx.sync()
    print("metric %s ChromaMEEnable %d: %.2f ms per 1080p frame (search range %d)" % (metric, cme, (time.perf_counter() - t0) * 1e3, R), flush=True)
ctx.close()
