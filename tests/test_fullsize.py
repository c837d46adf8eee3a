"""BASELINE configs 2 and 4 at full size (1920x1088 and 3840x2160, FullSearch +-32, 41 partitions per macroblock) on the device:
 * EVERY macroblock of the frame against the oracle's C driver, bit-exact (the oracle runs on the box's host cores in parallel);
 * the whole frame through size-independent properties: the two independent integer-search kernels (pair-lane "fast" path
   and the union-window generic path, forced by masking one partition off) must agree on every other partition of every
   macroblock; a resident re-run reproduces the first run; costs are consistent with the vectors they come with."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle
from tests.test_me import lambda_factors

pytestmark = pytest.mark.gpu
W, H, R = 1920, 1088, 32


def clip(W=W, H=H):
    rng = np.random.default_rng(20260410)
    B = rng.integers(0, 256, (H // 8 + 16, W // 8 + 16)).astype(np.float64)
    B = np.kron(B, np.ones((8, 8)))
    k = np.ones(9) / 9.0
    B = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, B)
    B = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, B)
    ref = np.clip(np.round(B[32:32 + H, 32:32 + W] + rng.normal(0, 2, (H, W))), 0, 255).astype(np.uint8)
    cur = np.clip(np.round(B[32 - 3:32 - 3 + H, 32 + 5:32 + 5 + W] + rng.normal(0, 2, (H, W))), 0, 255).astype(np.uint8)
    return cur, ref


@pytest.mark.parametrize("W,H", [(1920, 1088), (3840, 2160)])          # BASELINE configs 2 and 4
def test_full_frame_search(pkg, W, H):
    from h264_amd.jmhip import ME_MB_DTYPE
    cur, ref = clip(W, H)
    mbw, mbh = W // 16, H // 16
    n = mbw * mbh
    rng = np.random.default_rng(5)
    mbs = np.zeros(n, dtype=ME_MB_DTYPE)
    mbs["mb_x"], mbs["mb_y"], mbs["ref_is_0"] = np.arange(n) % mbw, np.arange(n) // mbw, 1
    mbs["pred_mv"] = (np.array([16, -16]) + rng.integers(-8, 9, (n, 1, 2))) + np.zeros((n, 41, 2), int)
    ctx = pkg.Context(W, H, yuv_format=0, max_refs=1, search_range=R)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    lam = lambda_factors(28)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    got = ctx.me_frame(prm, mbs)
    ctx.me_frame_async(prm, None, n)
    again = ctx.me_results(n)
    prm.partition_mask = ((1 << 41) - 1) & ~(1 << 40)          # forces the generic union-window kernel
    generic = ctx.me_frame(prm, mbs)
    ctx.close()

    for key in ("mv_int", "cost_int", "mv", "cost"):
        assert np.array_equal(got[key], again[key]), "resident re-run differs: " + key
        assert np.array_equal(got[key][:, :40], generic[key][:, :40]), "pair-lane and generic kernels disagree: " + key
    # a quarter-pel result lies within 3 quarter-pels of the integer one and never costs more than the carried SATD chain allows
    d = got["mv"].astype(int) - 4 * got["mv_int"].astype(int)
    assert np.abs(d).max() <= 3
    assert (np.abs(got["mv_int"].astype(int) - np.trunc(mbs["pred_mv"] / 4).astype(int)) <= R).all()
    # the clip moves by (5,-3): the 16x16 vector must find it on the bulk of the frame
    hit = ((got["mv_int"][:, 0, 0] == 5) & (got["mv_int"][:, 0, 1] == -3)).mean()
    assert hit > 0.9, hit

    # ---- the oracle over the WHOLE frame, macroblock rows dealt to host threads (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    L = oracle.lib()
    p = oracle.me_params(rdopt=1)
    rp = oracle.RefPic(ref, yuv_format=0)
    lam_a = (C.c_int * 3)(*lam)
    cur16 = np.ascontiguousarray(cur, dtype=np.uint16)
    mv_out = np.zeros((n, 41, 2), np.int16)
    cost_out = np.zeros((n, 41), np.int32)
    vp = C.c_void_p
    L.jmo_hotpath_mbs.restype = C.c_longlong
    L.jmo_hotpath_mbs.argtypes = [C.POINTER(oracle.MeParams), C.POINTER(oracle.Ref), vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), vp, vp, vp, vp]
    xy = np.ascontiguousarray(np.stack([mbs["mb_x"], mbs["mb_y"]], 1).astype(np.int16))
    preds = np.ascontiguousarray(mbs["pred_mv"].astype(np.int16))

    def rows(r):
        a, b = r * mbw, (r + 1) * mbw
        L.jmo_hotpath_mbs(C.byref(p), C.byref(rp.ref), cur16.ctypes.data, None, None, W, xy[a:b].ctypes.data, preds[a:b].ctypes.data, b - a, R, lam_a,
                          None, None, mv_out[a:b].ctypes.data, cost_out[a:b].ctypes.data)
    with ThreadPoolExecutor(max_workers=14) as ex:
        list(ex.map(rows, range(mbh)))
    assert np.array_equal(got["mv"], mv_out), "vectors differ from the oracle: %d macroblocks" % int((got["mv"] != mv_out).any(axis=(1, 2)).sum())
    assert np.array_equal(got["cost"], cost_out), "costs differ from the oracle"


@pytest.mark.parametrize("W,H,step", [(1920, 1088, 1), (3840, 2160, 9)])
def test_full_frame_residual_stage(pkg, W, H, step):
    """The frame stage (MC -> residual -> dct_4x4 / dct_chroma -> thresholds -> recon) over every macroblock of a 1080p / 2160p 4:2:0
    frame against the reference assembled from the oracle: ALL 8160 macroblocks at 1080p, every 9th (3600, all rows and columns hit) at 2160p;
    plus: the reconstruction of a second identical run is identical."""
    from h264_amd.jmhip import ME_MB_DTYPE
    cur, ref = clip(W, H)
    mk = lambda img, s: np.clip(np.round(128 + s * 0.25 * (img[::2, ::2].astype(float) - 128)), 0, 255).astype(np.uint8)
    curs, refs = (cur, mk(cur, 1), mk(cur, -1)), (ref, mk(ref, 1), mk(ref, -1))
    mbw, mbh = W // 16, H // 16
    n = mbw * mbh
    rng = np.random.default_rng(9)
    mbs = np.zeros(n, dtype=ME_MB_DTYPE)
    mbs["mb_x"], mbs["mb_y"], mbs["ref_is_0"] = np.arange(n) % mbw, np.arange(n) // mbw, 1
    mbs["pred_mv"] = (np.array([16, -16]) + rng.integers(-8, 9, (n, 1, 2))) + np.zeros((n, 41, 2), int)
    ctx = pkg.Context(W, H, yuv_format=1, max_refs=1, search_range=R)
    ctx.ref_upload(0, *refs)
    ctx.interp_luma(0)
    ctx.interp_chroma(0)
    ctx.cur_upload(*curs)
    lam = lambda_factors(28)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = 0, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    me = ctx.me_frame(prm, mbs)
    quants = np.array([pkg.flat_quant(28 + d, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1) for d in (0, 0, 3)], dtype=pkg.QUANT_DTYPE)
    ctx.residual_frame(quants, None)
    got = ctx.residual_download(n)
    recon = ctx.recon_download()
    ctx.residual_frame(quants, None)
    recon2 = ctx.recon_download()
    ctx.close()
    for a, b in zip(recon, recon2):
        assert np.array_equal(a, b)
    assert (got["cbp"] & 15).max() > 0 and ((got["cbp"] & 15) == 0).any()       # coded and uncoded macroblocks both occur

    sel = np.arange(0, n, step)
    rp = oracle.RefPic(refs[0], refs[1], refs[2], yuv_format=1)
    want = oracle.residual_frame(rp, curs, mbs[sel], me["mv"][sel], got["modes"][sel], quants, pkg.TQ_JOB_DTYPE, yuv_format=1)
    assert np.array_equal(got["cbp"][sel], want["cbp"])
    assert np.array_equal(got["cbp_blk"][sel], want["cbp_blk"])

    def lists_differ(gl, gr, wl, wr):
        """(level, run) lists compared up to and including the terminating zero level (what JM's entropy coder reads), vectorised over all rows"""
        w = wl.reshape(-1, wl.shape[-1])
        g, r_g, r_w = gl.reshape(w.shape), gr.reshape(w.shape), wr.reshape(w.shape)
        k = np.where((w == 0).any(axis=1), np.argmax(w == 0, axis=1), w.shape[1] - 1)[:, None]
        idx = np.arange(w.shape[1])[None, :]
        return bool((np.where(idx <= k, g != w, False)).any() or (np.where(idx < k, r_g != r_w, False)).any())
    csel = np.stack([2 * sel, 2 * sel + 1], axis=1).reshape(-1)
    assert not lists_differ(got["luma"]["levels"][sel], got["luma"]["runs"][sel], want["luma"]["levels"], want["luma"]["runs"]), "luma levels / runs"
    assert not lists_differ(got["chroma"]["levels"][csel][:, :4, :16], got["chroma"]["runs"][csel][:, :4, :16], want["chroma"]["levels"][:, :4, :16], want["chroma"]["runs"][:, :4, :16]), "chroma AC levels / runs"
    assert not lists_differ(got["chroma"]["dc_levels"][csel], got["chroma"]["dc_runs"][csel], want["chroma"]["dc_levels"], want["chroma"]["dc_runs"]), "chroma DC levels / runs"
    assert np.array_equal(got["luma"]["coeff_cost"][sel], want["luma"]["coeff_cost"])
    assert (want["luma"]["levels"][:, :, 0] != 0).any()       # coefficients do occur (this clip's chroma is flat enough to quantise to nothing)
    for i in sel:
        x, y = int(mbs[i]["mb_x"]) * 16, int(mbs[i]["mb_y"]) * 16
        assert np.array_equal(recon[0][y:y + 16, x:x + 16], want["recon"][0][y:y + 16, x:x + 16]), ("luma recon", i)
        assert np.array_equal(recon[1][y // 2:y // 2 + 8, x // 2:x // 2 + 8], want["recon"][1][y // 2:y // 2 + 8, x // 2:x // 2 + 8]), ("Cb recon", i)
        assert np.array_equal(recon[2][y // 2:y // 2 + 8, x // 2:x // 2 + 8], want["recon"][2][y // 2:y // 2 + 8, x // 2:x // 2 + 8]), ("Cr recon", i)
