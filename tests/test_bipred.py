"""Bi-predictive search of the 16x16 block (FullPelBlockMotionBiPred me_fullsearch.c:164, SubPelBlockSearchBiPred :520,
computeBiPredSAD1/2 + computeBiPredSATD1/2 me_distortion.c:482-1040) against the oracle, bit-exact. The oracle's
restatement is pinned inside the real JM by tests/test_oracle_swap.py (mask 0x100)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle
from tests.test_me import lambda_factors, make_pair


class Bipred(C.Structure):
    _fields_ = [("ref1", C.POINTER(oracle.Ref)), ("ref2", C.POINTER(oracle.Ref)), ("umv1", C.c_int), ("umv2", C.c_int),
                ("test8x8", C.c_int), ("max_val", C.c_int), ("apply_weights", C.c_int), ("weight1", C.c_int), ("weight2", C.c_int),
                ("offset_bi", C.c_int), ("wp_luma_round", C.c_int), ("luma_log_weight_denom", C.c_int), ("metric", C.c_int * 3),
                ("start_hp", C.c_int), ("start_qp", C.c_int)]


def oracle_bipred(rp1, rp2, cur16, job, lam, t8x8, wp):
    L = oracle.lib()
    b = Bipred()
    b.ref1, b.ref2 = C.pointer(rp1.ref), C.pointer(rp2.ref)
    b.test8x8, b.max_val = t8x8, 255
    b.apply_weights = 1 if wp else 0
    if wp:
        b.weight1, b.weight2, b.offset_bi, b.wp_luma_round, b.luma_log_weight_denom = wp
    b.metric[0], b.metric[1], b.metric[2] = 0, 2, 2
    b.start_hp, b.start_qp = 0, 1
    ox, oy = int(job["mb_x"]) * 16, int(job["mb_y"]) * 16
    orig = np.zeros(768, np.uint16)
    orig[:256] = cur16[oy:oy + 16, ox:ox + 16].reshape(-1)
    mv = np.array(job["mv"], np.int16).copy()
    smv = np.array(job["s_mv"], np.int16).copy()
    vp, ip = C.c_void_p, C.c_int
    if int(job["stage"]) == 0:
        L.jmo_fullpel_bipred.argtypes = [C.POINTER(Bipred), vp, ip, ip, ip, ip, ip, ip, ip, vp, vp, vp, vp, ip, ip, ip]
        cost = L.jmo_fullpel_bipred(C.byref(b), orig.ctypes.data, ox, oy, 1, int(job["pred1"][0]), int(job["pred1"][1]),
                                    int(job["pred2"][0]), int(job["pred2"][1]), mv.ctypes.data, mv[1:].ctypes.data,
                                    smv.ctypes.data, smv[1:].ctypes.data, int(job["search_range"]), int(job["min_mcost"]), lam[0])
    else:
        L.jmo_subpel_bipred.argtypes = [C.POINTER(Bipred), vp, ip, ip, ip, ip, ip, vp, vp, vp, vp, ip, ip, ip, vp]
        lam_a = (C.c_int * 3)(*lam)
        cost = L.jmo_subpel_bipred(C.byref(b), orig.ctypes.data, ox, oy, 1, int(job["pred2"][0]), int(job["pred2"][1]),
                                   mv.ctypes.data, mv[1:].ctypes.data, smv.ctypes.data, smv[1:].ctypes.data, 9, 9, int(job["min_mcost"]), lam_a)
    return int(mv[0]), int(mv[1]), cost


@pytest.mark.gpu
@pytest.mark.parametrize("t8x8,wp", [(0, None), (1, None), (0, (37, 29, -3, 16, 5)), (1, (64, 64, 2, 32, 6)), (0, (1, 1, 0, 0, 0))])
def test_bipred_search(pkg, t8x8, wp):
    from h264_amd.jmhip import BIPRED_JOB_DTYPE
    rng = np.random.default_rng(7 + t8x8)
    w, h = 96, 64
    cur, ref1 = make_pair(rng, w, h, "shift")
    _, ref2 = make_pair(rng, w, h, "shift")
    ref2 = np.roll(ref2, (1, -2), (0, 1))
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=2, search_range=16)
    for s, r in enumerate((ref1, ref2)):
        ctx.ref_upload(s, r)
        ctx.interp_luma(s)
    ctx.cur_upload(cur)
    lam = lambda_factors(30)
    prm = pkg.BipredParams()
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.transform8x8_mode = t8x8
    if wp:
        prm.apply_weights = 1
        prm.weight1, prm.weight2, prm.offset_bi, prm.wp_luma_round, prm.luma_log_weight_denom = wp
    mbw, mbh = w // 16, h // 16
    jobs = np.zeros(2 * mbw * mbh + 8, dtype=BIPRED_JOB_DTYPE)
    for i in range(len(jobs)):
        j = jobs[i]
        j["mb_x"], j["mb_y"] = (i // 2) % mbw, ((i // 2) // mbw) % mbh
        j["ref1"], j["ref2"] = (0, 1) if i % 3 else (1, 0)
        j["stage"] = i % 2
        far = 30 if i >= 2 * mbw * mbh else 6          # the last jobs push both blocks outside the picture (UMV)
        if j["stage"] == 0:
            j["s_mv"] = rng.integers(-far, far + 1, 2)
            j["mv"] = rng.integers(-far, far + 1, 2)
            j["search_range"] = [16, 8, 4, 2][i % 4]
            j["min_mcost"] = 2147483647 if i % 5 else 1500   # a carried minimum most candidates cannot beat
        else:
            j["s_mv"] = rng.integers(-4 * far, 4 * far + 1, 2)
            j["mv"] = 4 * rng.integers(-far, far + 1, 2)
            j["min_mcost"] = 2147483647
        j["pred1"] = rng.integers(-12, 13, 2)
        j["pred2"] = rng.integers(-12, 13, 2)
    got = ctx.bipred_search(prm, jobs)
    ctx.close()
    rp = [oracle.RefPic(ref1, yuv_format=0), oracle.RefPic(ref2, yuv_format=0)]
    cur16 = cur.astype(np.uint16)
    for i, j in enumerate(jobs):
        want = oracle_bipred(rp[int(j["ref1"])], rp[int(j["ref2"])], cur16, j, lam, t8x8, wp)
        assert (int(got[i]["mv"][0]), int(got[i]["mv"][1]), int(got[i]["cost"])) == want, (i, j)
