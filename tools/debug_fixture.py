"""Development aid: first picture of a real-JM field fixture through the device and the oracle; prints the first macroblocks whose records differ."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle
from tests.conftest import load_pkg
from tests.test_golden_field import CASES, GOLD, replay
name = sys.argv[1] if len(sys.argv) > 1 else "field_epzs_r16_2ref"
pkg = load_pkg()
case = CASES[name] if isinstance(CASES[name], tuple) else (CASES[name], 0, 1)
mode, t8, cavlc = case[:3]
z = np.load(os.path.join(GOLD, name + ".npz"))
W, H = int(z["f0_head"][0]), int(z["f0_head"][1])
R = 16
ctx = pkg.Context(W, H, yuv_format=0, max_refs=2, search_range=R)
ctx.slice_state_reset()
epzs = oracle.Epzs(W, H, R, 2) if mode == 3 else None
lib = pkg.load_library()
want = replay(name, CASES[name])
k = 0
head = z["f%d_head" % k]; nref = int(head[3]); refinfo = z["f%d_refinfo" % k]
for r in range(nref):
    ctx.ref_upload(r, z["f%d_refs" % k][r]); ctx.interp_luma(r)
ctx.cur_upload(z["f%d_cur" % k])
p = pkg.slice_host.slice_params(mode, R, nref, [int(v) for v in head[6:9]], int(head[9]), W, H=H, t8=t8, cavlc=cavlc)
if mode == 3:
    ids = (refinfo[:, 1].astype(np.int64) & 0xffffffff) | (refinfo[:, 2].astype(np.int64) << 32)
    epzs.slice_init(int(head[10]), [int(v) for v in refinfo[:, 0]], ids, z["f%d_col_mv" % k], z["f%d_col_ref_id" % k], num_ref_idx_l0_active=int(head[11]))
    ctx.epzs_colocated_upload(oracle.epzs_colocated(epzs, W, H))
    lib.jmhip_epzs_scales(p, int(head[10]), (C.c_int * nref)(*[int(v) for v in refinfo[:, 0]]), nref)
got = ctx.p_slice_search(p)
rec = want[0][1]
shown = 0
for i in range(len(got)):
    bad = [f for f in got.dtype.names if not np.array_equal(got[f][i], rec[f][i])]
    if bad:
        print("mb", i, "differs in", bad)
        for f in bad[:6]:
            print("   ", f, "device", np.asarray(got[f][i]).ravel()[:24].tolist(), "oracle", np.asarray(rec[f][i]).ravel()[:24].tolist())
        shown += 1
        if shown >= 3:
            break
print("mb0 device best_mode %d min_cost %d, oracle best_mode %d min_cost %d" % (got["best_mode"][0], got["min_cost"][0], rec["best_mode"][0], rec["min_cost"][0]))
print("sweeps", ctx.slice_passes())
