"""jmhip_distortion_batch (computeSAD / computeSADWP / computeSATD / computeSATDWP, src/me_distortion.c:351-800) and
jmhip_me_subpel (SubPelBlockMotionSearch alone) against the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle
from tests.test_me import lambda_factors, make_mbs, make_pair

pytestmark = pytest.mark.gpu


def oracle_dist(rp, cur16, job):
    L = oracle.lib()
    d = oracle.Dist()
    d.ref = C.pointer(rp.ref)
    d.umv, d.chroma_me, d.test8x8 = int(job["umv"]), 0, int(job["use_satd"] == 2)
    d.max_val, d.max_val_uv = 255, 255
    d.weight_luma, d.offset_luma = int(job["weight"]), int(job["offset"])
    d.wp_luma_round, d.luma_log_weight_denom = int(job["wp_round"]), int(job["wp_denom"])
    bsx, bsy = int(job["bsx"]), int(job["bsy"])
    px, py = int(job["pic_x"]), int(job["pic_y"])
    src = np.zeros(768, np.uint16)
    src[: bsx * bsy] = cur16[py:py + bsy, px:px + bsx].reshape(-1)
    fn = {(0, 0): L.jmo_sad, (0, 1): L.jmo_sad_wp, (1, 0): L.jmo_satd, (1, 1): L.jmo_satd_wp}[(int(job["use_satd"] > 0), int(job["wp"]))]
    fn.argtypes = [C.POINTER(oracle.Dist), C.c_void_p] + [C.c_int] * 5
    # JM's candidate coordinates are relative to the padded plane origin: cand = (pic + IMG_PAD_SIZE)*4 + mv
    return fn(C.byref(d), src.ctypes.data, bsy, bsx, 2147483647, int(job["cand_x"]), int(job["cand_y"]))


@pytest.mark.parametrize("umv", [0, 1])
def test_distortion_batch(pkg, umv):
    from h264_amd.jmhip import DIST_JOB_DTYPE
    rng = np.random.default_rng(11 + umv)
    w, h = 96, 64
    cur, ref = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=16)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    n = 600
    jobs = np.zeros(n, dtype=DIST_JOB_DTYPE)
    sizes = [4, 8, 16]
    for i in range(n):
        j = jobs[i]
        j["bsx"], j["bsy"] = sizes[rng.integers(3)], sizes[rng.integers(3)]
        j["pic_x"] = 4 * rng.integers(0, (w - j["bsx"]) // 4 + 1)
        j["pic_y"] = 4 * rng.integers(0, (h - j["bsy"]) // 4 + 1)
        sat = int(rng.integers(3))
        if sat == 2 and ((j["bsx"] | j["bsy"]) & 7):
            sat = 1
        j["use_satd"] = sat
        if umv:   # anywhere, far outside included: JM clamps the block origin (per SATD sub-block)
            j["cand_x"] = (j["pic_x"] + 20) * 4 + rng.integers(-4 * (w + 40), 4 * (w + 40))
            j["cand_y"] = (j["pic_y"] + 20) * 4 + rng.integers(-4 * (h + 40), 4 * (h + 40))
        else:     # FAST_ACCESS: block inside the padded plane
            j["cand_x"] = rng.integers(0, 4 * (w + 40 - j["bsx"]) + 1)
            j["cand_y"] = rng.integers(0, 4 * (h + 40 - j["bsy"]) + 1)
        j["umv"] = umv
        j["wp"] = int(rng.integers(2))
        j["wp_denom"] = int(rng.integers(0, 8))
        j["wp_round"] = (1 << (j["wp_denom"] - 1)) if j["wp_denom"] else 0
        j["weight"] = int(rng.integers(-64, 128))
        j["offset"] = int(rng.integers(-30, 31))
    got = ctx.distortion_batch(jobs)
    ctx.close()
    rp = oracle.RefPic(ref, yuv_format=0)
    cur16 = cur.astype(np.uint16)
    want = np.array([oracle_dist(rp, cur16, jobs[i]) for i in range(n)], np.int32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]], jobs[bad[:5]])


def test_distortion_batch_rejects_bad_jobs(pkg):
    from h264_amd.jmhip import DIST_JOB_DTYPE
    ctx = pkg.Context(64, 48, yuv_format=0, max_refs=1, search_range=16)
    ref = np.zeros((48, 64), np.uint8)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(ref)
    j = np.zeros(1, dtype=DIST_JOB_DTYPE)
    j["bsx"], j["bsy"] = 16, 16
    j["cand_x"], j["cand_y"] = 4 * 200, 0           # FAST access beyond the plane: refused, not wrapped
    with pytest.raises(pkg.JmhipError):
        ctx.distortion_batch(j)
    j["cand_x"] = 0
    j["bsx"] = 12
    with pytest.raises(pkg.JmhipError):
        ctx.distortion_batch(j)
    ctx.close()


@pytest.mark.parametrize("rdopt,t8x8", [(1, 0), (0, 0), (1, 1)])
def test_me_subpel_alone(pkg, rdopt, t8x8):
    """Sub-pel refinement from ARBITRARY integer vectors (not only the integer search's own answer)."""
    rng = np.random.default_rng(5)
    w, h, R = 64, 48, 8
    cur, ref = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=R)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, 12)
    lam = lambda_factors(28)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt, prm.is_b_slice = -1, R, rdopt, 0
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.transform8x8_mode, prm.subpel, prm.partition_mask = t8x8, 1, (1 << 41) - 1
    from h264_amd.jmhip import ME_RESULT_DTYPE
    res = np.zeros(len(mbs), dtype=ME_RESULT_DTYPE)
    res["mv_int"] = rng.integers(-14, 15, (len(mbs), 41, 2))
    got = ctx.me_subpel(prm, mbs, res)
    ctx.close()

    L = oracle._setup_search_protos()
    p = oracle.me_params(rdopt=rdopt, transform8x8_mode=t8x8)
    rp = oracle.RefPic(ref, yuv_format=0)
    cur16 = cur.astype(np.uint16)
    lam_a = (C.c_int * 3)(*lam)
    for i, mb in enumerate(mbs):
        ox, oy = int(mb["mb_x"]) * 16, int(mb["mb_y"]) * 16
        for q, (bt, x4, y4, w4, h4) in enumerate(oracle.PARTS):
            px, py, bsx, bsy = ox + 4 * x4, oy + 4 * y4, 4 * w4, 4 * h4
            orig = np.zeros(768, np.uint16)
            orig[: bsx * bsy] = cur16[py:py + bsy, px:px + bsx].reshape(-1)
            mvq = (res["mv_int"][i, q].astype(np.int16) * 4).copy()
            cost = L.jmo_subpel_search(C.byref(p), C.byref(rp.ref), orig.ctypes.data, 1, px, py, bt,
                                       int(mb["pred_mv"][q][0]), int(mb["pred_mv"][q][1]),
                                       mvq.ctypes.data, mvq[1:].ctypes.data, 9, 9, 2147483647, lam_a)
            assert (int(got["mv"][i, q, 0]), int(got["mv"][i, q, 1]), int(got["cost"][i, q])) == (int(mvq[0]), int(mvq[1]), cost), (i, q)


@pytest.mark.parametrize("wp", [None, (37, -6, 16, 5)])
def test_distortion_surface(pkg, wp):
    """Row-segment SADs and per-block SATDs of every integer displacement, against the oracle's computeSAD / computeSATD
    evaluated on exactly those pieces (1x4 rows, 4x4 and 8x8 blocks) under UMV access."""
    from h264_amd.jmhip import SURFACE_JOB_DTYPE
    rng = np.random.default_rng(23)
    w, h, R = 64, 48, 4
    cur, ref = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=16)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    jobs = np.zeros(3, dtype=SURFACE_JOB_DTYPE)
    # an inner macroblock, the top-left one pushed out of the picture, the bottom-right one pushed far out
    for i, (mx, my, cx, cy) in enumerate([(1, 1, 2, -1), (0, 0, -14, -9), (3, 2, 30, 25)]):
        jobs[i]["mb_x"], jobs[i]["mb_y"], jobs[i]["R"], jobs[i]["cx"], jobs[i]["cy"] = mx, my, R, cx, cy
        if wp:
            jobs[i]["wp"], jobs[i]["weight"], jobs[i]["offset"], jobs[i]["wp_round"], jobs[i]["wp_denom"] = (1,) + wp
    sad = ctx.distortion_surface("sad_rows", jobs)
    satd = ctx.distortion_surface("satd_blocks", jobs)
    ctx.close()

    L = oracle.lib()
    rp = oracle.RefPic(ref, yuv_format=0)
    cur16 = cur.astype(np.uint16)
    proto = [C.POINTER(oracle.Dist), C.c_void_p] + [C.c_int] * 5
    L.jmo_sad.argtypes = proto
    L.jmo_satd.argtypes = proto

    def dist(test8x8):
        d = oracle.Dist()
        d.ref = C.pointer(rp.ref)
        d.umv, d.chroma_me, d.test8x8, d.max_val, d.max_val_uv = 1, 0, test8x8, 255, 255
        if wp:
            d.weight_luma, d.offset_luma, d.wp_luma_round, d.luma_log_weight_denom = wp
        return d
    d4, d8 = dist(0), dist(1)
    L.jmo_sad_wp.argtypes = proto
    L.jmo_satd_wp.argtypes = proto
    f_sad, f_satd = (L.jmo_sad_wp, L.jmo_satd_wp) if wp else (L.jmo_sad, L.jmo_satd)
    for i, job in enumerate(jobs):
        ox, oy = int(job["mb_x"]) * 16, int(job["mb_y"]) * 16
        for ay in range(2 * R + 1):
            for ax in range(2 * R + 1):
                mvx, mvy = int(job["cx"]) - R + ax, int(job["cy"]) - R + ay
                for r in range(16):
                    for g in range(4):
                        src = np.zeros(768, np.uint16)
                        src[:4] = cur16[oy + r, ox + 4 * g:ox + 4 * g + 4]
                        want = f_sad(C.byref(d4), src.ctypes.data, 1, 4, 2147483647, (ox + 4 * g + 20 + mvx) * 4, (oy + r + 20 + mvy) * 4)
                        assert int(sad[i, ay, ax, r, g]) == want, (i, ay, ax, r, g)
                for b in range(16):
                    bx, by = 4 * (b & 3), 4 * (b >> 2)
                    src = np.zeros(768, np.uint16)
                    src[:16] = cur16[oy + by:oy + by + 4, ox + bx:ox + bx + 4].reshape(-1)
                    want = f_satd(C.byref(d4), src.ctypes.data, 4, 4, 2147483647, (ox + bx + 20 + mvx) * 4, (oy + by + 20 + mvy) * 4)
                    assert int(satd[i, ay, ax, b]) == want, (i, ay, ax, b)
                for b in range(4):
                    bx, by = 8 * (b & 1), 8 * (b >> 1)
                    src = np.zeros(768, np.uint16)
                    src[:64] = cur16[oy + by:oy + by + 8, ox + bx:ox + bx + 8].reshape(-1)
                    want = f_satd(C.byref(d8), src.ctypes.data, 8, 8, 2147483647, (ox + bx + 20 + mvx) * 4, (oy + by + 20 + mvy) * 4)
                    assert int(satd[i, ay, ax, 16 + b]) == want, (i, ay, ax, "8x8", b)
