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
    jobs["blocks"] = 0xffff
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


@pytest.mark.parametrize("weighted", [0, 1])
def test_pred_cost_bipred_weights_and_partition_masks(pkg, weighted):
    """p_dir 2 and weighted predictions (LumaPrediction macroblock.c:836-945) and per-partition block masks (BIDPartitionCost
    mv-search.c:1050): the prediction is formed here from the oracle's planes with JM's formulas, the costs by the oracle."""
    from h264_amd.jmhip import PREDCOST_JOB_DTYPE
    rng = np.random.default_rng(77 + weighted)
    w, h = 96, 64
    cur, ref0 = make_pair(rng, w, h, "shift")
    _, ref1 = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=2, search_range=8)
    for s, r in enumerate((ref0, ref1)):
        ctx.ref_upload(s, r)
        ctx.interp_luma(s)
    ctx.cur_upload(cur)
    mbw, mbh = w // 16, h // 16
    masks = [0xffff, 0x00ff, 0xff00, 0x3333, 0xcccc, 0x0033, 0x00cc, 0x3300, 0xcc00]      # 16x16, 16x8 x2, 8x16 x2, the four 8x8
    jobs = np.zeros(4 * mbw * mbh, dtype=PREDCOST_JOB_DTYPE)
    for i in range(len(jobs)):
        j = jobs[i]
        j["mb_x"], j["mb_y"] = i % mbw, (i // mbw) % mbh
        j["blocks"] = masks[i % len(masks)]
        far = 4 * 50 if i % 5 == 0 else 12
        j["mv"][:] = rng.integers(-far, far + 1, 2)
        j["mv1"][:] = rng.integers(-far, far + 1, 2)
        j["ref"][:], j["ref1"][:] = i % 2, 1 - i % 2
        j["bi"][:] = 1 if i % 3 else 0
        if i % 7 == 0:                                  # mixed macroblock: per-block direction
            j["bi"] = rng.integers(0, 2, 16)
        j["weighted"], j["wp_denom"] = weighted, 5
        j["wp_round"] = 16
        j["w0"], j["w1"], j["off"] = rng.integers(-20, 80, 16), rng.integers(-20, 80, 16), rng.integers(-12, 13, 16)
    got4 = ctx.pred_cost_batch(jobs, metric=2, layout=1)
    got0 = ctx.pred_cost_batch(jobs, metric=0, layout=1)
    ctx.close()

    L = oracle.lib()
    L.jmo_pred_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.jmo_pred_costs.restype = None
    L.jmo_hadamard_sad4x4.argtypes = [C.c_void_p]
    L.jmo_hadamard_sad8x8.argtypes = [C.c_void_p]
    rps = [oracle.RefPic(ref0, yuv_format=0), oracle.RefPic(ref1, yuv_format=0)]
    Wp, Hp = w + 40, h + 40

    def fetch(slot, ox, oy, x4, y4, mv):
        xq, yq = ((ox + 4 * x4 + 20) << 2) + int(mv[0]), ((oy + 4 * y4 + 20) << 2) + int(mv[1])
        xpos, ypos = min(max(xq >> 2, 0), Wp - 17), min(max(yq >> 2, 0), Hp - 17)
        return rps[slot].luma[yq & 3, xq & 3, ypos:ypos + 4, xpos:xpos + 4].astype(np.int64)

    for i, j in enumerate(jobs):
        ox, oy = int(j["mb_x"]) * 16, int(j["mb_y"]) * 16
        diff = np.zeros((16, 16), np.int32)
        for b in range(16):
            x4, y4 = b & 3, b >> 2
            a = fetch(int(j["ref"][b]), ox, oy, x4, y4, j["mv"][b])
            if j["bi"][b]:
                q = fetch(int(j["ref1"][b]), ox, oy, x4, y4, j["mv1"][b])
                if weighted:
                    p = np.clip(((int(j["w0"][b]) * a + int(j["w1"][b]) * q + 2 * 16) >> 6) + int(j["off"][b]), 0, 255)
                else:
                    p = (a + q + 1) >> 1
            else:
                p = np.clip(((int(j["w0"][b]) * a + 16) >> 5) + int(j["off"][b]), 0, 255) if weighted else a
            diff[4 * y4:4 * y4 + 4, 4 * x4:4 * x4 + 4] = cur[oy + 4 * y4:oy + 4 * y4 + 4, ox + 4 * x4:ox + 4 * x4 + 4].astype(np.int64) - p
        for metric, got in ((2, got4), (0, got0)):
            c4 = c8 = 0
            for b in range(16):
                if (int(j["blocks"]) >> b) & 1:
                    d = np.ascontiguousarray(diff[4 * (b >> 2):4 * (b >> 2) + 4, 4 * (b & 3):4 * (b & 3) + 4].reshape(-1), dtype=np.int32)
                    c4 += L.jmo_hadamard_sad4x4(d.ctypes.data) if metric == 2 else int(np.abs(d).sum())
            for b8 in range(4):
                o = 8 * (b8 >> 1) + 2 * (b8 & 1)
                need = (1 << o) | (1 << (o + 1)) | (1 << (o + 4)) | (1 << (o + 5))
                if (int(j["blocks"]) & need) == need:
                    d = np.ascontiguousarray(diff[8 * (b8 >> 1):8 * (b8 >> 1) + 8, 8 * (b8 & 1):8 * (b8 & 1) + 8].reshape(-1), dtype=np.int32)
                    c8 += L.jmo_hadamard_sad8x8(d.ctypes.data) if metric == 2 else int(np.abs(d).sum())
            assert (int(got[i, 0]), int(got[i, 1])) == (c4, c8), (i, metric, hex(int(j["blocks"])))
