# times jmhip_me_frame at 1080p for predictor fields of increasing spread (distinct centres per MB)
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tests.conftest import load_pkg
pkg = load_pkg()
from h264_amd.jmhip import ME_MB_DTYPE
W, H, R = 1920, 1088, 32
rng = np.random.default_rng(0)
ref = rng.integers(0, 256, (H, W), dtype=np.uint8)
cur = np.roll(ref, (2, -3), (0, 1))
ctx = pkg.Context(W, H, yuv_format=0, max_refs=1, search_range=R)
ctx.ref_upload(0, ref); ctx.interp_luma(0); ctx.cur_upload(cur)
n = (W // 16) * (H // 16)
prm = pkg.MeParams()
prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
prm.level_mv_min, prm.level_mv_max = -511, 511
prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = 1000000
prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
for spread in (0, 3, 4, 5, 6, 8):
    mbs = np.zeros(n, dtype=ME_MB_DTYPE)
    mbs["mb_x"] = np.arange(n) % (W // 16); mbs["mb_y"] = np.arange(n) // (W // 16); mbs["ref_is_0"] = 1
    base = rng.integers(-8, 9, (n, 1, 2))
    mbs["pred_mv"] = base + rng.integers(-spread, spread + 1, (n, 41, 2)) if spread else base + np.zeros((n, 41, 2), int)
    cent = (np.trunc(mbs["pred_mv"] / 4)).astype(int)
    ng = np.mean([len({(a, b) for a, b in c}) for c in cent[:500]])
    ctx.me_frame(prm, mbs)
    ctx.timing_enable(True); ctx.timing_read()
    for _ in range(5): ctx.me_frame_async(prm, None, n)
    ctx.sync()
    t = ctx.timing_read()
    print("spread", spread, "distinct centres/MB %.1f" % ng, "me_int %.3f ms  me_sub %.3f ms" % (t["me_int"][0] / max(t["me_int"][1], 1), t["me_sub"][0] / max(t["me_sub"][1], 1)), flush=True)
ctx.close()
