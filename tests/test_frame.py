"""Frame stage (MC prediction -> residual -> dct_4x4/dct_chroma -> coefficient-cost thresholds -> recon) vs a reference
assembled from the oracle's planes and dct_* functions."""
import numpy as np
import pytest

from tests import oracle
from tests.test_me import lambda_factors, make_mbs
from tests.test_tq import compare_lists


def synth(rng, w, h, fmt):
    yy, xx = np.mgrid[0:h + 32, 0:w + 32]
    base = ((np.sin(xx / 6.0) * np.cos(yy / 9.0)) * 70 + 128 + rng.normal(0, 10, (h + 32, w + 32))).clip(0, 255)
    ref = base[16:16 + h, 16:16 + w].astype(np.uint8)
    cur = (base[13:13 + h, 21:21 + w] + rng.normal(0, 2, (h, w))).clip(0, 255).astype(np.uint8)
    hc, wc = (h // 2, w // 2) if fmt == 1 else (h, w // 2)
    sub = (slice(None, None, 2), slice(None, None, 2)) if fmt == 1 else (slice(None), slice(None, None, 2))
    mk = lambda img, s: np.clip(np.round(128 + s * 0.25 * (img[sub].astype(float) - 128)), 0, 255).astype(np.uint8)[:hc, :wc]
    return (cur, mk(cur, 1), mk(cur, -1)), (ref, mk(ref, 1), mk(ref, -1))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,qp,given_modes,chroma_planes", [(1, 28, False, True), (1, 40, True, True), (2, 24, False, True), (1, 12, False, True),
                                                              (1, 28, False, False), (1, 40, True, False), (2, 24, False, False)])
def test_residual_frame(pkg, fmt, qp, given_modes, chroma_planes):
    """chroma_planes False: jmhip_interp_chroma is not called and the frame stage computes the eighth-pel chroma samples itself;
    the expectation (the oracle's planes) is the same."""
    rng = np.random.default_rng(qp + fmt)
    w, h, R = 64, 48, 8
    cur, ref = synth(rng, w, h, fmt)
    ctx = pkg.Context(w, h, yuv_format=fmt, max_refs=1, search_range=R)
    ctx.ref_upload(0, *ref)
    ctx.interp_luma(0)
    if chroma_planes:
        ctx.interp_chroma(0)
    ctx.cur_upload(*cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, 6)
    lam = lambda_factors(qp)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    me = ctx.me_frame(prm, mbs)
    qpc = qp                                            # QP_SCALE_CR is the identity below 30; larger qp: any table works for parity
    quants = np.array([pkg.flat_quant(qp, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1),
                       pkg.flat_quant(qpc, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1),
                       pkg.flat_quant(qpc + 3, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1)], dtype=pkg.QUANT_DTYPE)
    modes = None
    if given_modes:
        modes = np.zeros(len(mbs), dtype=pkg.MB_MODE_DTYPE)
        modes["mode"] = rng.choice([1, 2, 3, 8], len(mbs))
        modes["b8mode"] = rng.integers(4, 8, (len(mbs), 4))
    if fmt == 1:
        ctx.frame_keep_prediction()
    ctx.residual_frame(quants, modes)
    got = ctx.residual_download(len(mbs))
    recon = ctx.recon_download()
    records, pred = (ctx.residual_records(len(mbs)), ctx.pred_download()) if fmt == 1 else (None, None)
    if fmt != 1:                                        # the dense records and the prediction picture are the fused 4:2:0 stage's
        with pytest.raises(pkg.JmhipError):
            ctx.residual_records(len(mbs))
        with pytest.raises(pkg.JmhipError):
            ctx.pred_download()
    ctx.close()

    want_modes = modes if given_modes else oracle.pick_modes(me["cost"])
    assert np.array_equal(got["modes"]["mode"], want_modes["mode"])
    p8 = got["modes"]["mode"] == 8
    assert np.array_equal(got["modes"]["b8mode"][p8] if given_modes else got["modes"]["b8mode"], want_modes["b8mode"][p8] if given_modes else want_modes["b8mode"])
    rp = oracle.RefPic(ref[0], ref[1], ref[2], yuv_format=fmt)
    want = oracle.residual_frame(rp, cur, mbs, me["mv"], got["modes"], quants, pkg.TQ_JOB_DTYPE, yuv_format=fmt)
    compare_lists(got["luma"]["levels"], got["luma"]["runs"], want["luma"]["levels"], want["luma"]["runs"], "luma")
    nb = 4 if fmt == 1 else 8
    compare_lists(got["chroma"]["levels"][:, :nb, :16], got["chroma"]["runs"][:, :nb, :16], want["chroma"]["levels"][:, :nb, :16], want["chroma"]["runs"][:, :nb, :16], "chroma AC")
    compare_lists(got["chroma"]["dc_levels"][:, None], got["chroma"]["dc_runs"][:, None], want["chroma"]["dc_levels"][:, None], want["chroma"]["dc_runs"][:, None], "chroma DC")
    assert np.array_equal(got["luma"]["coeff_cost"], want["luma"]["coeff_cost"])
    assert np.array_equal(got["cbp"], want["cbp"])
    assert np.array_equal(got["cbp_blk"], want["cbp_blk"])
    for g, wv, name in zip(recon, want["recon"], "YUV"):
        assert np.array_equal(g, wv), "recon %s" % name
    if records is not None:
        # jmhip_residual_records_download: the same results as dense jmhip_mb_residual records (what a slice-level binding answers JM's dct_4x4 /
        # dct_chroma from), and jmhip_pred_download: img->mpr of every macroblock
        for i, mb in enumerate(mbs):
            r, x, y = records[i], 16 * int(mb["mb_x"]), 16 * int(mb["mb_y"])
            for b in range(16):
                n = int(r["cnt"][b])
                assert np.array_equal(r["lev"][b, :n], want["luma"]["levels"][i, b, :n]) and np.array_equal(r["run"][b, :n], want["luma"]["runs"][i, b, :n]) and want["luma"]["levels"][i, b, n] == 0
                assert ((int(r["nonzero"]) >> b) & 1) == int(want["luma"]["nonzero"][i, b]) and int(r["coeff_cost"][b]) == int(want["luma"]["coeff_cost"][i, b])
            assert np.array_equal(r["recon_y"], want["luma"]["recon"][i]) and np.array_equal(r["fadj_y"], want["luma"]["fadjust"][i])
            for uv in range(2):
                wc = {k: v[2 * i + uv] for k, v in want["chroma"].items()}
                for b in range(4):
                    n = int(r["cnt"][16 + 4 * uv + b])
                    assert np.array_equal(np.where(r["ac_zeroed"][uv], 0, r["lev"][16 + 4 * uv + b, :n]), wc["levels"][b, :n]) and np.array_equal(r["run"][16 + 4 * uv + b, :n], wc["runs"][b, :n])
                n = int(r["dc_cnt"][uv])
                assert np.array_equal(r["dc_lev"][uv, :n], wc["dc_levels"][:n]) and np.array_equal(r["dc_run"][uv, :n], wc["dc_runs"][:n]) and wc["dc_levels"][n] == 0
                assert int(r["ret"][uv]) == int(wc["ret"]) and int(r["cbp_blk"][uv]) == int(wc["cbp_blk"]) and int(r["cbp_clear"][uv]) == int(wc["cbp_clear"])
                assert np.array_equal(r["recon_c"][uv], wc["recon"][:8, :8])
                assert np.array_equal(pred[1 + uv][y // 2:y // 2 + 8, x // 2:x // 2 + 8], want["jobs_c"][2 * i + uv]["pred"][:8, :8]), "chroma prediction of macroblock %d" % i
            assert np.array_equal(pred[0][y:y + 16, x:x + 16], want["jobs_y"][i]["pred"]), "luma prediction of macroblock %d" % i
    # the thresholds must actually be exercised somewhere across the parametrisation
    assert got["cbp"].max() > 0 or qp >= 40


@pytest.mark.gpu
@pytest.mark.parametrize("cavlc,qp,far,chroma_planes", [(1, 28, 6, True), (0, 22, 6, True), (1, 34, 45, True), (1, 34, 45, False), (0, 26, 90, False)])
def test_residual_frame_with_8x8_transform_macroblocks(pkg, cavlc, qp, far, chroma_planes):
    """A mix of 4x4- and 8x8-transform macroblocks (luma_transform_size_8x8_flag per mode): dct_8x8 branch of
    LumaResidualCoding8x8, 8x8-granular prediction clamp (far = vectors beyond the padded plane), CAVLC interleave / CABAC lists."""
    rng = np.random.default_rng(qp + cavlc)
    w, h, R = 64, 48, 8
    cur, ref = synth(rng, w, h, 1)
    ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    ctx.ref_upload(0, *ref)
    ctx.interp_luma(0)
    if chroma_planes:
        ctx.interp_chroma(0)                           # without them: vectors far beyond the picture exercise the on-the-fly clamps
    ctx.cur_upload(*cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, 4 * far)
    lam = lambda_factors(qp)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.transform8x8_mode, prm.subpel, prm.partition_mask = 1, 1, (1 << 41) - 1
    me = ctx.me_frame(prm, mbs)
    q8 = pkg.flat_quant(qp, 342, is8x8=True, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=cavlc, transform8x8_flag=1)
    quants = np.array([pkg.flat_quant(qp, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=cavlc),
                       pkg.flat_quant(qp, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=cavlc),
                       pkg.flat_quant(qp + 3, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=cavlc), q8], dtype=pkg.QUANT_DTYPE)
    modes = np.zeros(len(mbs), dtype=pkg.MB_MODE_DTYPE)
    modes["mode"] = rng.choice([1, 2, 3, 8], len(mbs))
    modes["b8mode"] = rng.integers(4, 8, (len(mbs), 4))
    t8 = rng.integers(0, 2, len(mbs)).astype(bool)
    t8[0] = True
    modes["b8mode"][t8] = 4                               # the 8x8 transform needs partitions of at least 8x8
    modes["pad"][:, 0] = t8
    ctx.residual_frame(quants, modes)
    got = ctx.residual_download(len(mbs))
    recon = ctx.recon_download()
    bad = modes.copy()
    bad["b8mode"][0] = 7
    bad["mode"][0] = 8
    with pytest.raises(pkg.JmhipError):
        ctx.residual_frame(quants, bad)                   # 8x8 transform over 4x4 partitions: refused
    ctx.close()

    rp = oracle.RefPic(ref[0], ref[1], ref[2], yuv_format=1)
    want = oracle.residual_frame(rp, cur, mbs, me["mv"], modes, quants, pkg.TQ_JOB_DTYPE, yuv_format=1)
    for key in ("levels", "runs", "levels8", "runs8"):
        a, b = got["luma"][key], want["luma"][key]
        if key in ("levels", "runs"):
            compare_lists(got["luma"]["levels"], got["luma"]["runs"], want["luma"]["levels"], want["luma"]["runs"], "luma 4x4 lists")
        else:
            compare_lists(got["luma"]["levels8"][t8], got["luma"]["runs8"][t8], want["luma"]["levels8"][t8], want["luma"]["runs8"][t8], "luma 8x8 lists")
    assert np.array_equal(got["cbp"], want["cbp"])
    assert np.array_equal(got["cbp_blk"], want["cbp_blk"])
    for g, wv, name in zip(recon, want["recon"], "YUV"):
        assert np.array_equal(g, wv), "recon %s" % name
    assert (got["cbp"][t8] & 15).max() > 0 or qp >= 34


@pytest.mark.gpu
@pytest.mark.parametrize("weighted,planes,fused", [(False, False, True), (True, True, True), (True, False, False), (False, True, False)])
def test_b_macroblocks_second_list(pkg, weighted, planes, fused, monkeypatch):
    """jmhip_frame_bipred_set: per 8x8 block list 0 only / list 1 only / both, list-1 vectors per 4x4 block from a second reference slot,
    plain average or JM's weighted mixes (incl. the luma denominator in the bi-predictive chroma shift) -- reconstruction and cbp of every
    macroblock against the oracle, through the fused kernel and through the separate kernels, chroma from planes and computed."""
    if not fused:
        monkeypatch.setenv("JMHIP_FRAME_FUSED", "0")
    rng = np.random.default_rng(31 + weighted + 2 * planes)
    W, H, R = 96, 64, 8
    n = (W // 16) * (H // 16)
    base = rng.integers(0, 256, (H + 32, W + 32)).astype(np.float64)
    for _ in range(2):
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 0)) / 4
    def frame(dx, dy):
        Y = np.clip(base[16 + dy:16 + dy + H, 16 + dx:16 + dx + W] + rng.normal(0, 2, (H, W)), 0, 255).astype(np.uint8)
        U = np.clip(Y[::2, ::2].astype(int) // 2 + 60, 0, 255).astype(np.uint8)
        V = np.clip(220 - Y[1::2, 1::2].astype(int) // 2, 0, 255).astype(np.uint8)
        return Y, U, V
    ref0, ref1, cur = frame(0, 0), frame(4, -3), frame(2, -1)
    ctx = pkg.Context(W, H, yuv_format=1, max_refs=2, search_range=R)
    for slot, ref in ((0, ref0), (1, ref1)):
        ctx.ref_upload(slot, *ref)
        ctx.interp_luma(slot)
        if planes:
            ctx.interp_chroma(slot)
    ctx.cur_upload(*cur)
    mbs = np.zeros(n, dtype=pkg.ME_MB_DTYPE)
    for i in range(n):
        mbs[i]["mb_x"], mbs[i]["mb_y"], mbs[i]["ref"], mbs[i]["ref_is_0"] = i % (W // 16), i // (W // 16), 0, 1
        mbs[i]["pred_mv"][:] = rng.integers(-6, 7, 2)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = 0, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = 30000
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    me = ctx.me_frame(prm, mbs)
    bi = np.zeros(n, dtype=pkg.MB_BIPRED_DTYPE)
    bi["pdir"] = rng.integers(0, 3, (n, 4))
    bi["ref1"] = 1
    bi["mv1"] = rng.integers(-40, 41, (n, 16, 2))
    bi["mv1"][0] = [[-300, -200]] * 16                     # far outside the picture: the UMV clamps
    wp = bw = None
    if weighted:
        wp = {"luma_round": 16, "luma_denom": 5, "chroma_round": 4, "chroma_denom": 3, "weight": np.zeros((16, 3), int), "offset": np.zeros((16, 3), int)}
        wp["weight"][0], wp["offset"][0] = (30, 9, 7), (3, -2, 1)
        bw = {"w0": rng.integers(10, 40, (4, 4, 3)), "w1": rng.integers(10, 40, (4, 4, 3)), "weight1": rng.integers(5, 12, (4, 3)), "offset1": rng.integers(-4, 5, (4, 3))}
        bw["weight1"][:, 0] += 22
    ctx.frame_wp_set(wp)
    ctx.frame_bipred_set(bi, bw)
    quants = np.array([pkg.flat_quant(26 + d, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1) for d in (0, 0, 3)], dtype=pkg.QUANT_DTYPE)
    ctx.residual_frame(quants, None)
    got = ctx.residual_download(n)
    recon = ctx.recon_download()
    ctx.frame_bipred_set(None)
    ctx.frame_wp_set(None)
    ctx.close()
    assert (bi["pdir"] == 0).any() and (bi["pdir"] == 1).any() and (bi["pdir"] == 2).any()
    rps = [oracle.RefPic(*ref0, yuv_format=1), oracle.RefPic(*ref1, yuv_format=1)]
    want = oracle.residual_frame(rps, cur, mbs, me["mv"], got["modes"], quants, pkg.TQ_JOB_DTYPE, yuv_format=1, blk_ref=np.zeros((n, 4), int), wp=wp, bi=bi, bw=bw)
    assert np.array_equal(got["cbp"], want["cbp"]) and np.array_equal(got["cbp_blk"], want["cbp_blk"])
    for k in range(3):
        assert np.array_equal(recon[k], want["recon"][k]), "plane %d" % k
    assert (got["cbp"] != 0).any()
