"""jmhip_me_frame / jmhip_me_subpel with every error metric of JM's computeUniPred dispatch (MEDistortionFPel / HPel / QPel = SAD, SSE,
Hadamard SAD; mv-search.c:400-424) and the chroma term (ChromaMEEnable 1 / 2, me_distortion.c:376-402), against the oracle:
bit-exact (mv_int, cost_int, mv, cost) for all 41 partitions. me_metric.hip is the kernel behind metric_set = 1."""
import numpy as np
import pytest

from tests import oracle
from tests.test_me import lambda_factors, make_pair, make_mbs


def make_chroma(rng, w, h, kind, yuv_format):
    cw, ch = (w // 2 if yuv_format in (1, 2) else w), (h // 2 if yuv_format == 1 else h)
    if kind == "flat":
        mk = lambda v: np.full((ch, cw), v, np.uint8)
        return (mk(90), mk(160)), (mk(90), mk(160))
    yy, xx = np.mgrid[0:ch + 16, 0:cw + 16]
    out = []
    for k in range(2):
        base = ((np.sin(xx / (4.0 + k)) * np.cos(yy / (5.0 - k))) * 60 + 128 + rng.normal(0, 10, (ch + 16, cw + 16))).clip(0, 255)
        ref = base[8:8 + ch, 8:8 + cw].astype(np.uint8)
        cur = (base[8 - 1:8 - 1 + ch, 8 + 2:8 + 2 + cw] + rng.normal(0, 2, (ch, cw))).clip(0, 255).astype(np.uint8)
        if kind == "noise":
            ref = rng.integers(0, 256, (ch, cw)).astype(np.uint8)
            cur = rng.integers(0, 256, (ch, cw)).astype(np.uint8)
        out.append((cur, ref))
    return (out[0][0], out[1][0]), (out[0][1], out[1][1])


def run_case(pkg, w, h, kind, mode, R, rdopt, spread, metric, chroma_me=0, chroma_w=1, t8x8=0, per_partition=True, subpel=1, seed=0,
             mask=(1 << 41) - 1, is_b=0, wp=None, yuv_format=1, qp=28, split=False):
    rng = np.random.default_rng(seed)
    cur, ref = make_pair(rng, w, h, kind)
    cur_uv, ref_uv = make_chroma(rng, w, h, kind, yuv_format)
    ctx = pkg.Context(w, h, yuv_format=yuv_format, max_refs=1, search_range=R)
    ctx.ref_upload(0, ref, ref_uv[0], ref_uv[1])
    ctx.interp_luma(0)
    ctx.interp_chroma(0)
    ctx.cur_upload(cur, cur_uv[0], cur_uv[1])
    mbs = make_mbs(pkg, rng, w // 16, h // 16, spread, per_partition)
    lam = lambda_factors(qp)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt, prm.is_b_slice = mode, R, rdopt, is_b
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.transform8x8_mode, prm.subpel, prm.partition_mask = t8x8, subpel, mask
    prm.metric_set = 1
    prm.metric[0], prm.metric[1], prm.metric[2] = metric
    prm.chroma_me, prm.chroma_me_weight = chroma_me, chroma_w
    if wp:                                           # (luma weight, offset, denominator, (cb weight, offset), (cr weight, offset))
        prm.wp_enable, prm.wp_denom, prm.wp_round = 1, wp[2], (1 << (wp[2] - 1)) if wp[2] else 0
        prm.wp_weight[0], prm.wp_offset[0] = wp[0], wp[1]
        prm.wp_chroma_denom, prm.wp_chroma_round = wp[2], (1 << (wp[2] - 1)) if wp[2] else 0
        for k in range(2):
            prm.wp_weight_cr[0][k], prm.wp_offset_cr[0][k] = wp[3 + k]
    if split:                                        # integer stage and refinement as two calls (the per-call binding of the JM shim)
        prm.subpel = 0
        got = ctx.me_frame(prm, mbs)
        prm.subpel = 1
        got = ctx.me_subpel(prm, mbs, got)
    else:
        got = ctx.me_frame(prm, mbs)
    ctx.close()

    p = oracle.me_params(rdopt=rdopt, is_b_slice=is_b, transform8x8_mode=t8x8, metric=metric, chroma_me=chroma_me, chroma_me_weight=chroma_w)
    if wp:
        p.apply_weights, p.weight_luma, p.offset_luma = 1, wp[0], wp[1]
        p.luma_log_weight_denom, p.wp_luma_round = wp[2], (1 << (wp[2] - 1)) if wp[2] else 0
        p.chroma_log_weight_denom, p.wp_chroma_round = wp[2], (1 << (wp[2] - 1)) if wp[2] else 0
        for k in range(2):
            p.weight_cr[k], p.offset_cr[k] = wp[3 + k]
    refpic = oracle.RefPic(ref, ref_uv[0], ref_uv[1], yuv_format=yuv_format)
    want = oracle.me_frame(p, [refpic], cur, mbs, mode, R, lam, subpel=bool(subpel), mask=mask, cur_uv=cur_uv if chroma_me else None)
    for key in ("mv_int", "cost_int", "mv", "cost"):
        g, wv = got[key], want[key]
        if not np.array_equal(g, wv):
            bad = np.argwhere(g != wv)[0]
            raise AssertionError("%s differs at mb %d partition %d: got %s want %s (pred %s; mv_int got %s want %s)" % (
                key, bad[0], bad[1], g[bad[0], bad[1]], wv[bad[0], bad[1]], mbs[bad[0]]["pred_mv"][bad[1]],
                got["mv_int"][bad[0], bad[1]], want["mv_int"][bad[0], bad[1]]))


METRICS = [(1, 1, 1), (2, 2, 2), (0, 0, 0), (1, 2, 2), (2, 0, 1), (0, 1, 2), (0, 2, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("mode,rdopt", [(-1, 1), (-1, 0), (0, 0)])
def test_metrics_luma(pkg, metric, mode, rdopt):
    """SAD / SSE / Hadamard SAD at each level, FullSearch and FastFullSearch (which takes squared error for every metric but SAD),
    rdopt off: the zero-vector bonuses and, at macroblock (0, 0), the wrapped early-exit bound of spiral position 0."""
    run_case(pkg, 64, 48, "shift", mode, 8, rdopt, 6, metric, seed=sum(metric) * 3 + mode + 2)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,t8x8", [((2, 2, 2), 1), ((2, 0, 2), 1), ((1, 2, 2), 1)])
def test_metrics_with_the_8x8_hadamard(pkg, metric, t8x8):
    """Transform8x8Mode: block types 1..4 take HadamardSAD8x8 wherever the level's metric is Hadamard SAD -- at integer positions too."""
    run_case(pkg, 64, 48, "shift", -1, 8, 1, 6, metric, t8x8=t8x8, seed=11)
    run_case(pkg, 64, 48, "noise", 0, 8, 0, 30, metric, t8x8=t8x8, seed=12)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,chroma_me,chroma_w,mode", [
    ((0, 2, 2), 1, 1, -1), ((0, 2, 2), 2, 1, -1), ((0, 0, 0), 2, 2, -1), ((1, 1, 1), 2, 1, -1), ((1, 0, 2), 1, 3, -1),
    ((0, 2, 2), 1, 1, 0), ((1, 1, 0), 2, 2, 0), ((2, 2, 2), 2, 1, -1), ((0, 0, 2), 2, 1, 0)])
def test_chroma_term(pkg, metric, chroma_me, chroma_w, mode):
    """ChromaMEEnable: the Cb / Cr blocks join SAD and SSE (never Hadamard SAD) at integer positions (1) and at sub-pel positions too
    (2), weighted by ChromaMEWeight -- except in SetupFastFullPelSearch, which adds them unweighted; ChromaMEEnable = 1 also makes both
    refinements start from scratch."""
    run_case(pkg, 64, 48, "shift", mode, 8, 1, 6, metric, chroma_me=chroma_me, chroma_w=chroma_w, seed=5 + chroma_w)
    run_case(pkg, 64, 48, "shift", mode, 8, 0, 4, metric, chroma_me=chroma_me, chroma_w=chroma_w, seed=6 + chroma_w, per_partition=False)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,chroma_me,mode", [((0, 2, 2), 2, -1), ((1, 1, 1), 2, -1), ((0, 0, 0), 2, 0), ((2, 2, 2), 0, -1)])
def test_windows_outside_the_picture(pkg, metric, chroma_me, mode):
    """Far predictors on noise: blocks leave the picture and the padding ring (UMV access: luma and chroma origins clamped); flat
    pictures: every candidate ties, the spiral order decides."""
    run_case(pkg, 64, 48, "noise", mode, 8, 1, 60, metric, chroma_me=chroma_me, seed=21)
    run_case(pkg, 64, 48, "flat", mode, 8, 0, 4, metric, chroma_me=chroma_me, seed=22)


@pytest.mark.gpu
@pytest.mark.parametrize("yuv_format", [2, 3])
def test_chroma_term_422_444(pkg, yuv_format):
    run_case(pkg, 64, 48, "shift", -1, 8, 1, 6, (0, 0, 2), chroma_me=2, chroma_w=1, seed=31, yuv_format=yuv_format)
    run_case(pkg, 64, 48, "noise", 0, 8, 0, 40, (1, 1, 1), chroma_me=2, chroma_w=2, seed=32, yuv_format=yuv_format)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,chroma_me,mode", [((0, 0, 2), 2, -1), ((2, 2, 2), 0, -1), ((0, 2, 2), 1, 0), ((2, 2, 0), 0, 0), ((1, 1, 1), 2, -1), ((1, 0, 1), 1, 0)])
def test_weighted_reference(pkg, metric, chroma_me, mode):
    """UseWeightedReferenceME with the chroma term: computeSADWP weights Cb / Cr with their own weight and offset (me_distortion.c:443-470)."""
    run_case(pkg, 64, 48, "shift", mode, 8, 1, 6, metric, chroma_me=chroma_me, seed=41, wp=(48, -9, 5, (40, 4), (27, -6)))


@pytest.mark.gpu
@pytest.mark.parametrize("metric,chroma_me", [((2, 2, 2), 0), ((0, 0, 0), 2), ((1, 1, 2), 0)])
def test_refinement_as_its_own_call(pkg, metric, chroma_me):
    """jmhip_me_subpel with metric_set: the integer result and -- when the integer and half-pel metric agree -- its cost are inputs."""
    run_case(pkg, 64, 48, "shift", -1, 8, 0, 6, metric, chroma_me=chroma_me, seed=51, split=True)


@pytest.mark.gpu
def test_range32_and_masks(pkg):
    run_case(pkg, 96, 64, "shift", -1, 32, 0, 3, (2, 2, 2), seed=61, per_partition=False, mask=0x1ff)
    run_case(pkg, 96, 64, "shift", 0, 32, 0, 3, (1, 1, 1), chroma_me=1, seed=62, mask=(1 << 41) - 1 - 0x1e)


@pytest.mark.gpu
def test_rejections(pkg):
    ctx = pkg.Context(64, 48, yuv_format=1, max_refs=1, search_range=8)
    rng = np.random.default_rng(1)
    cur, ref = make_pair(rng, 64, 48, "shift")
    ctx.ref_upload(0, ref, ref[:24, :32].copy(), ref[:24, :32].copy())
    ctx.interp_luma(0)
    ctx.cur_upload(cur, cur[:24, :32].copy(), cur[:24, :32].copy())
    mbs = make_mbs(pkg, rng, 4, 3, 4)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, 8, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = 4000
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    prm.metric_set = 1
    prm.metric[0], prm.metric[1], prm.metric[2] = 0, 2, 3
    with pytest.raises(pkg.JmhipError, match="metric"):
        ctx.me_frame(prm, mbs)
    prm.metric[2] = 2
    prm.chroma_me = 1
    with pytest.raises(pkg.JmhipError, match="jmhip_interp_chroma"):
        ctx.me_frame(prm, mbs)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,R", [(-1, 48), (0, 64)])
def test_search_ranges_beyond_the_packed_keys(pkg, mode, R):
    """search_range > 44: the spiral index no longer fits the fast kernels' 13-bit tie field; the general kernel (64-bit keys) takes the
    search, with JM's default metrics (metric_set = 0)."""
    from tests.test_me import run_case as run_default
    run_default(pkg, 96, 64, "shift", mode, R, 0, 3, per_partition=False, seed=R)
