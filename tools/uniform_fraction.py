"""How many macroblocks of the bench clip end the integer search with ONE vector for all 41 partitions (diagnostic for me_sub's uniform phase)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import __graft_entry__ as ge
pkg = ge._load_pkg()
frames = bench.synth_frames(3)
W, H = frames[0][0].shape[1], frames[0][0].shape[0]
ctx = pkg.Context(W, H, yuv_format=1, max_refs=1, search_range=32)
ctx.ref_upload(0, *frames[0])
ctx.interp_luma(0)
ctx.cur_upload(*frames[1])
mbs = bench.make_jobs(pkg, list(range(H // 16)))
prm = pkg.MeParams()
prm.search_mode, prm.search_range, prm.rdopt = -1, 32, 1
prm.level_mv_min, prm.level_mv_max = -511, 511
prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = bench.lambda_factor(bench.QP)
prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
r = ctx.me_frame(prm, mbs)
mvi = r["mv_int"].astype(int)
uni = (mvi == mvi[:, :1]).all(axis=(1, 2))
mvq = r["mv"].astype(int)
uniq = (mvq == mvq[:, :1]).all(axis=(1, 2))
distinct = np.array([len({(a, b) for a, b in m}) for m in mvi])
print("macroblocks with one integer vector for all 41 partitions: %.1f %%; one final vector: %.1f %%; distinct integer vectors per macroblock: mean %.2f" % (
    100 * uni.mean(), 100 * uniq.mean(), distinct.mean()))
ctx.close()
