"""Motion estimation: jmhip_me_frame (integer full search + sub-pel refinement) vs the oracle, bit-exact on
(mv_int, cost_int, mv, cost) for all 41 partitions."""
import numpy as np
import pytest

from tests import oracle


def lambda_factors(qp=28):
    """SetLagrangianMultipliers for P slices, rdopt on (slice.c:1329-1358): lambda_mf = LAMBDA_FACTOR(sqrt(0.85*2^((qp-12)/3)))."""
    lam = int(65536 * np.sqrt(0.85 * 2 ** ((qp - 12) / 3.0)) + 0.5)
    return [lam, lam, lam]


def make_pair(rng, w, h, kind):
    yy, xx = np.mgrid[0:h + 32, 0:w + 32]
    base = ((np.sin(xx / 6.0) * np.cos(yy / 9.0)) * 70 + 128 + rng.normal(0, 12, (h + 32, w + 32))).clip(0, 255)
    if kind == "noise":
        ref = rng.integers(0, 256, (h, w)).astype(np.uint8)
        cur = rng.integers(0, 256, (h, w)).astype(np.uint8)
    elif kind == "flat":      # every candidate ties: exercises the spiral tie-break
        ref = np.full((h, w), 100, np.uint8)
        cur = np.full((h, w), 100, np.uint8)
    else:                     # translation (5,-3) + noise, like the synthetic clip of SURVEY 8(d)
        ref = base[16:16 + h, 16:16 + w].astype(np.uint8)
        cur = (base[16 - 3:16 - 3 + h, 16 + 5:16 + 5 + w] + rng.normal(0, 2, (h, w))).clip(0, 255).astype(np.uint8)
    return cur, ref


def make_mbs(pkg, rng, mbw, mbh, spread, per_partition=True):
    from h264_amd.jmhip import ME_MB_DTYPE
    mbs = np.zeros(mbw * mbh, dtype=ME_MB_DTYPE)
    for i in range(mbw * mbh):
        mbs[i]["mb_x"], mbs[i]["mb_y"] = i % mbw, i // mbw
        mbs[i]["ref"], mbs[i]["ref_is_0"] = 0, 1
        if per_partition:
            mbs[i]["pred_mv"] = rng.integers(-spread, spread + 1, (41, 2))
        else:
            mbs[i]["pred_mv"][:] = rng.integers(-spread, spread + 1, 2)
    return mbs


def run_case(pkg, w, h, kind, mode, R, rdopt, spread, t8x8=0, per_partition=True, subpel=1, seed=0, mask=(1 << 41) - 1, is_b=0, wp=None):
    rng = np.random.default_rng(seed)
    cur, ref = make_pair(rng, w, h, kind)
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=R)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, spread, per_partition)
    lam = lambda_factors(28)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt, prm.is_b_slice = mode, R, rdopt, is_b
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.transform8x8_mode, prm.subpel, prm.partition_mask = t8x8, subpel, mask
    if wp:                                           # (weight, offset, denominator) of the one reference slot
        prm.wp_enable, prm.wp_denom, prm.wp_round = 1, wp[2], (1 << (wp[2] - 1)) if wp[2] else 0
        prm.wp_weight[0], prm.wp_offset[0] = wp[0], wp[1]
    got = ctx.me_frame(prm, mbs)
    ctx.close()

    p = oracle.me_params(rdopt=rdopt, is_b_slice=is_b, transform8x8_mode=t8x8)
    if wp:
        p.apply_weights, p.weight_luma, p.offset_luma = 1, wp[0], wp[1]
        p.luma_log_weight_denom, p.wp_luma_round = wp[2], (1 << (wp[2] - 1)) if wp[2] else 0
    want = oracle.me_frame(p, [oracle.RefPic(ref, yuv_format=0)], cur, mbs, mode, R, lam, subpel=bool(subpel), mask=mask)
    for key in ("mv_int", "cost_int", "mv", "cost"):
        g, wv = got[key], want[key]
        if not np.array_equal(g, wv):
            bad = np.argwhere(g != wv)[0]
            raise AssertionError("%s differs at mb %d partition %d: got %s want %s (pred %s)" % (
                key, bad[0], bad[1], g[bad[0], bad[1]], wv[bad[0], bad[1]], mbs[bad[0]]["pred_mv"][bad[1]]))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode,R,rdopt,spread,t8x8", [
    ("shift", -1, 8, 1, 8, 0),       # FullSearch, per-partition centres
    ("shift", -1, 8, 0, 8, 0),       # rdopt off: centre clip, check_for_00, check_position0
    ("shift", 0, 8, 1, 8, 0),        # FastFullSearch
    ("shift", 0, 8, 0, 8, 0),        # FastFullSearch + pos_00 pre-check
    ("noise", -1, 8, 1, 40, 0),      # far predictors: windows leave the picture (UMV), big union window
    ("noise", 0, 8, 0, 60, 1),       # 8x8 Hadamard in sub-pel
    ("flat", -1, 8, 0, 4, 0),        # all ties
    ("flat", 0, 8, 0, 4, 1),
    ("shift", -1, 16, 1, 6, 1),
])
def test_me_frame_small(pkg, kind, mode, R, rdopt, spread, t8x8):
    run_case(pkg, 64, 48, kind, mode, R, rdopt, spread, t8x8, seed=R * 7 + spread)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,R,spread", [(-1, 8, 0), (-1, 32, 2), (0, 32, 3)])
def test_b_slices_switch_the_zero_vector_bonuses_off(pkg, mode, R, spread):
    """img->type == B_SLICE: check_for_00 (me_fullsearch.c:75) and check_position0 (:361) do not apply even with rdopt off."""
    run_case(pkg, 96, 64, "shift", mode, R, 0, spread, per_partition=False, seed=5 + R, is_b=1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,R,spread,wp,t8x8", [(-1, 8, 6, (48, -9, 5), 0), (0, 8, 6, (23, 12, 5), 0), (-1, 32, 2, (70, -20, 6), 0), (0, 32, 3, (29, 5, 5), 1),
                                                   (0, 16, 3, (-12, 200, 4), 0), (-1, 8, 40, (40, 3, 5), 0)])
def test_weighted_reference_search(pkg, mode, R, spread, wp, t8x8):
    """UseWeightedReferenceME: every reference sample goes through clip(((w * p + round) >> denom) + o) before SAD (integer) and SATD
    (sub-pel) -- computeSADWP / computeSATDWP, the weighted line of the fast full search. All three integer kernels (generic for far
    predictors / R = 8, pair-lane for 2R+1 >= 32) and the 4x4 / 8x8 Hadamard paths; a negative weight; offsets that saturate."""
    run_case(pkg, 96, 64, "shift", mode, R, 1, spread, t8x8=t8x8, per_partition=(mode == -1), seed=R + wp[0], wp=wp)


@pytest.mark.gpu
def test_me_frame_qcif_range32(pkg):
    """BASELINE config 1/2 search range on a QCIF-sized picture (99 MBs x 41 partitions x 4225 candidates)."""
    run_case(pkg, 176, 144, "shift", -1, 32, 1, 8, seed=3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode,rdopt,per_partition,spread", [
    ("shift", 0, 1, True, 8),        # FastFullSearch: one centre per MB -> fast kernel
    ("shift", 0, 0, True, 8),        # + pos_00 pre-check
    ("shift", -1, 1, False, 8),      # FullSearch with one predictor per MB -> fast kernel
    ("shift", -1, 0, False, 8),      # + check_for_00 / check_position0
    ("flat", 0, 0, True, 4),         # every candidate ties: tie-break through the fast kernel's A/B spiral halves
    ("flat", -1, 0, False, 4),
    ("noise", 0, 1, True, 200),      # centre clipped by the +-2047 / level limits, windows far outside the picture
    ("noise", -1, 0, False, 100),
    ("shift", -1, 1, True, 3),       # FullSearch, 41 different predictors, ONE centre (pred/4 == 0): one fast item
    ("shift", -1, 1, True, 5),       # centres in {-1,0,1}^2: up to 9 distinct -> several fast items per MB, some MBs generic
    ("shift", -1, 0, True, 6),       # same with rdopt off (check_for_00 on the 16x16 item only)
    ("noise", -1, 1, True, 7),
])
def test_me_fast_kernel_range32(pkg, kind, mode, rdopt, per_partition, spread):
    run_case(pkg, 96, 64, kind, mode, 32, rdopt, spread, per_partition=per_partition, seed=spread + mode * 3 + rdopt)


@pytest.mark.gpu
@pytest.mark.parametrize("R,per_partition,spread", [(8, True, 0), (8, True, 3), (32, False, 0), (32, False, 2)])
def test_full_search_wrapped_early_exit_bound_at_picture_origin(pkg, R, per_partition, spread):
    """rdopt off, predictor within +-3 quarter-pels at macroblock (0,0): the check_for_00 bonus makes the first
    candidate's motion cost negative, JM's early-exit bound INT_MAX - mcost wraps, and computeSAD returns after one
    row (me_fullsearch.c:129-138). The real encoder does this on every P picture's first macroblock; the moving clip
    makes the one-row cost win against the true motion, so a device that computed the full SAD would be caught."""
    run_case(pkg, 96, 64, "shift", -1, R, 0, spread, per_partition=per_partition, seed=41 + spread)


@pytest.mark.gpu
@pytest.mark.parametrize("R,mode,rdopt,per_partition,spread", [(16, -1, 1, False, 8), (16, 0, 0, True, 8), (20, -1, 0, False, 30), (24, 0, 1, True, 60),
                                                             (31, -1, 1, True, 3), (40, 0, 0, True, 8)])
def test_me_pair_kernel_other_ranges(pkg, R, mode, rdopt, per_partition, spread):
    """The pair-lane kernel with one column group (32 <= 2R+1 < 64: four row bands + 1..31 remainder columns) and with the
    widest window (R = 40: 17 remainder columns)."""
    run_case(pkg, 96, 64, "shift", mode, R, rdopt, spread, per_partition=per_partition, seed=R + spread)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,rdopt,per_partition,spread", [(0, 1, True, 8), (-1, 0, False, 8), (-1, 1, True, 5), (0, 0, True, 4)])
def test_me_single_lane_kernel_still_agrees(pkg, mode, rdopt, per_partition, spread, monkeypatch):
    """JMHIP_ME_KERNEL=single: the one-lane-per-candidate kernel the pair-lane kernel replaced stays selectable and exact."""
    monkeypatch.setenv("JMHIP_ME_KERNEL", "single")
    run_case(pkg, 96, 64, "shift", mode, 32, rdopt, spread, per_partition=per_partition, seed=77 + spread)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,rdopt,per_partition,spread", [(0, 1, True, 8), (-1, 0, False, 8), (-1, 1, True, 5), (0, 0, True, 4)])
def test_me_persistent_kernel_one_item_per_workgroup(pkg, mode, rdopt, per_partition, spread, monkeypatch):
    """JMHIP_ME_KERNEL=pers: the persistent, double-buffered form of the pair-lane kernel (an experiment that measured slower; kept exact)."""
    monkeypatch.setenv("JMHIP_ME_KERNEL", "pers")
    run_case(pkg, 96, 64, "shift", mode, 32, rdopt, spread, per_partition=per_partition, seed=77 + spread)


@pytest.mark.gpu
@pytest.mark.parametrize("R,mode,rdopt,per_partition,spread,grid,size", [
    (32, 0, 0, True, 8, 8, (176, 144)), (32, -1, 0, True, 3, 8, (176, 144)), (32, -1, 1, False, 40, 16, (176, 144)), (16, 0, 1, True, 8, 8, (96, 64)),
    (40, -1, 0, True, 2, 8, (96, 64)), (24, 0, 0, True, 60, 24, (176, 144))])
def test_me_persistent_kernel_pipelines_many_items_per_workgroup(pkg, R, mode, rdopt, per_partition, spread, grid, size, monkeypatch):
    """me_int_pers_kernel with a small grid: every workgroup stages item k+1 while it walks item k and finishes item k-1 (dozens of items
    each, uneven counts per XCD run, windows hanging over every picture edge); bit-exact against the oracle like the one-item launches."""
    monkeypatch.setenv("JMHIP_ME_KERNEL", "pers")
    monkeypatch.setenv("JMHIP_ME_PERS_GRID", str(grid))
    run_case(pkg, size[0], size[1], "shift", mode, R, rdopt, spread, per_partition=per_partition, seed=R + spread + grid)


@pytest.mark.gpu
def test_me_mixed_fast_and_generic_macroblocks(pkg):
    """FullSearch where some MBs have one predictor (fast kernel) and others per-partition predictors (generic)."""
    rng = np.random.default_rng(5)
    w, h, R = 96, 64, 32
    cur, ref = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=R)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, 8, True)
    for i in range(0, len(mbs), 2):
        mbs[i]["pred_mv"][:] = mbs[i]["pred_mv"][0]
    lam = lambda_factors(28)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    got = ctx.me_frame(prm, mbs)
    got2 = None
    ctx.me_frame_async(prm, None, len(mbs))          # resident re-run must give the same answer
    got2 = ctx.me_results(len(mbs))
    ctx.close()
    p = oracle.me_params(rdopt=1)
    want = oracle.me_frame(p, [oracle.RefPic(ref, yuv_format=0)], cur, mbs, -1, R, lam)
    for key in ("mv_int", "cost_int", "mv", "cost"):
        assert np.array_equal(got[key], want[key]), key
        assert np.array_equal(got2[key], want[key]), key + " (resident)"


@pytest.mark.gpu
def test_me_integer_only_and_partition_mask(pkg):
    run_case(pkg, 64, 48, "shift", -1, 8, 1, 8, subpel=0, seed=11)
    run_case(pkg, 64, 48, "shift", -1, 8, 1, 8, mask=0b11111, seed=12)          # 16x16, 16x8, 8x16 only
    run_case(pkg, 64, 48, "shift", 0, 8, 0, 8, mask=((1 << 41) - 1) & ~1, seed=13)  # without the 16x16 partition


@pytest.mark.gpu
def test_partition_table_matches_jm_order(pkg):
    assert pkg.partition_table() == oracle.PARTS
