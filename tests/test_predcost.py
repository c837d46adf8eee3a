"""jmhip_pred_cost_batch: the RD-off mode-decision costs of TransformDecision (macroblock.c:1458, JM's sequential diff64 layout)
and GetSkipCostMB (mv-search.c:1136, raster layout) against the oracle's restatement, which is pinned inside the real JM
(tests/test_oracle_swap.py, mask 0x200)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle
from tests.test_me import make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", [2, 0])
@pytest.mark.parametrize("layout", [0, 1])
def test_pred_cost_batch(pkg, metric, layout):
    from h264_amd.jmhip import PREDCOST_JOB_DTYPE
    rng = np.random.default_rng(31 + metric + layout)
    w, h = 96, 64
    cur, ref0 = make_pair(rng, w, h, "shift")
    _, ref1 = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=2, search_range=8)
    for s, r in enumerate((ref0, ref1)):
        ctx.ref_upload(s, r)
        ctx.interp_luma(s)
    ctx.cur_upload(cur)
    mbw, mbh = w // 16, h // 16
    jobs = np.zeros(3 * mbw * mbh, dtype=PREDCOST_JOB_DTYPE)
    for i in range(len(jobs)):
        jobs[i]["mb_x"], jobs[i]["mb_y"] = i % mbw, (i // mbw) % mbh
        far = 4 * 60 if i >= 2 * mbw * mbh else 14      # the last third: vectors far beyond the padded plane (origin clamp per 4x4)
        if i % 2:
            jobs[i]["mv"] = rng.integers(-far, far + 1, (16, 2))          # 16 different vectors, two references
            jobs[i]["ref"] = rng.integers(0, 2, 16)
        else:
            jobs[i]["mv"][:] = rng.integers(-far, far + 1, 2)             # one vector (the skip-cost shape)
    got = ctx.pred_cost_batch(jobs, metric=metric, layout=layout)
    ctx.close()

    L = oracle.lib()
    L.jmo_pred_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.jmo_pred_costs.restype = None
    rps = [oracle.RefPic(ref0, yuv_format=0), oracle.RefPic(ref1, yuv_format=0)]
    Wp, Hp = w + 40, h + 40
    for i, job in enumerate(jobs):
        ox, oy = int(job["mb_x"]) * 16, int(job["mb_y"]) * 16
        mpr = np.zeros((16, 16), np.uint16)
        for b in range(16):
            x4, y4 = b & 3, b >> 2
            xq = ((ox + 4 * x4 + 20) << 2) + int(job["mv"][b][0])
            yq = ((oy + 4 * y4 + 20) << 2) + int(job["mv"][b][1])
            xpos, ypos = min(max(xq >> 2, 0), Wp - 17), min(max(yq >> 2, 0), Hp - 17)
            mpr[4 * y4:4 * y4 + 4, 4 * x4:4 * x4 + 4] = rps[int(job["ref"][b])].luma[yq & 3, xq & 3, ypos:ypos + 4, xpos:xpos + 4]
        cur16 = np.ascontiguousarray(cur[oy:oy + 16, ox:ox + 16], dtype=np.uint16)
        c4, c8 = C.c_int(), C.c_int()
        L.jmo_pred_costs(cur16.ctypes.data, mpr.ctypes.data, metric, layout, C.byref(c4), C.byref(c8))
        assert (int(got[i, 0]), int(got[i, 1])) == (c4.value, c8.value), (i, job["mv"][:2])
    # Note: the two diff64 layouts give the SAME sums. The sequential layout permutes the six index bits of the 8x8 block
    # ((a,r1,r0 | b,x1,x0) -> (a,b,r1 | r0,x1,x0)), and the 2-D 8-point Hadamard transform is the Walsh-Hadamard transform over
    # those six bits: a bit permutation only permutes its outputs, so the sum of magnitudes is unchanged (SAD trivially so).
    # JM's odd layout is therefore harmless; both are kept in the ABI and both are checked against the oracle here.
