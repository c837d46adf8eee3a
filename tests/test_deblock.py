"""In-loop deblocking filter on the device (jmhip_deblock_frame) against the oracle's restatement of DeblockFrame
(lencod/src/loopFilter.c:87), which is itself pinned inside the real JM (tests/test_oracle_swap.py). Bit-exact."""
import numpy as np
import pytest

from tests import oracle

INT64_MIN = -(1 << 63)


def make_case(pkg, rng, w, h, fmt, intra_frac=0.15, t8_frac=0.3, idc_mode="mixed", qp_lo=20, qp_hi=46, smooth=True, bslice=False, nslices=3):
    """A picture with blocking artefacts and the per-macroblock / per-block side information the filter reads."""
    mbw, mbh = w // 16, h // 16
    cw, ch = {0: (0, 0), 1: (w // 2, h // 2), 2: (w // 2, h), 3: (w, h)}[fmt]

    def plane(W, H):
        if not W:
            return None
        if smooth:      # smooth field + per-4x4-block offsets: many edges sit between the alpha / beta thresholds
            yy, xx = np.mgrid[0:H, 0:W]
            base = 128 + 60 * np.sin(xx / 37.0) * np.cos(yy / 29.0)
            blk = rng.integers(-9, 10, size=(H // 4 + 1, W // 4 + 1))
            img = base + np.kron(blk, np.ones((4, 4)))[:H, :W] + rng.integers(-2, 3, size=(H, W))
        else:
            img = rng.integers(116, 141, size=(H, W))       # flat noise: at high qp most lines pass the alpha / beta tests
        return np.clip(img, 0, 255).astype(np.uint8)

    Y, U, V = plane(w, h), plane(cw, ch), plane(cw, ch)
    mbs = np.zeros(mbw * mbh, pkg.DEBLOCK_MB_DTYPE)
    mbs["intra"] = rng.random(mbw * mbh) < intra_frac
    mbs["qp"] = rng.integers(qp_lo, qp_hi + 1, mbw * mbh)
    mbs["qpc"] = np.clip(mbs["qp"][:, None].astype(int) + rng.integers(-6, 3, (mbw * mbh, 2)), 0, 51)
    mbs["alpha_c0_offset"] = rng.integers(-6, 7) * 2 if idc_mode != "zero" else 0
    mbs["beta_offset"] = rng.integers(-6, 7) * 2 if idc_mode != "zero" else 0
    mbs["transform_8x8"] = rng.random(mbw * mbh) < t8_frac
    mbs["cbp_blk"] = rng.integers(0, 1 << 16, mbw * mbh) & rng.integers(0, 1 << 16, mbw * mbh)
    # slices of whole rows: availability as the encoder leaves it (same slice and inside the picture)
    slice_rows = max(1, -(-mbh // nslices))
    for i in range(mbw * mbh):
        x, y = i % mbw, i // mbw
        mbs["avail_a"][i] = x != 0
        mbs["avail_b"][i] = y != 0 and (y % slice_rows) != 0
    if idc_mode == "mixed":
        idc_of_slice = rng.integers(0, 3, mbh // slice_rows + 1)
        mbs["disable_idc"] = [idc_of_slice[(i // mbw) // slice_rows] for i in range(mbw * mbh)]
    elif idc_mode == "two":
        mbs["disable_idc"] = 2
    blks = np.zeros(16 * mbw * mbh, pkg.DEBLOCK_BLK_DTYPE)
    # vectors: per-macroblock base + small per-block jitter so that |dmv| sits on both sides of the limit 4
    mvb = rng.integers(-40, 41, (mbh, mbw, 2, 2))
    mv = np.kron(mvb.reshape(mbh, mbw, 4), np.ones((4, 4, 1), int)).reshape(mbh * 4, mbw * 4, 2, 2) + rng.integers(-3, 4, (mbh * 4, mbw * 4, 2, 2))
    blks["mv"] = mv.reshape(-1, 2, 2)
    ref0 = rng.integers(0, 3, 16 * mbw * mbh).astype(np.int64) * 2
    if bslice:
        ref1 = np.where(rng.random(16 * mbw * mbh) < 0.5, rng.integers(0, 3, 16 * mbw * mbh).astype(np.int64) * 2, INT64_MIN)
        ref0 = np.where(rng.random(16 * mbw * mbh) < 0.2, INT64_MIN, ref0)
    else:
        ref1 = np.full(16 * mbw * mbh, INT64_MIN, np.int64)
    blks["ref_id"][:, 0] = ref0
    blks["ref_id"][:, 1] = ref1
    return (Y, U, V), mbs, blks


def run(pkg, w, h, fmt, seed, **kw):
    rng = np.random.default_rng(seed)
    planes, mbs, blks = make_case(pkg, rng, w, h, fmt, **kw)
    want = oracle.deblock_frame(planes[0], planes[1], planes[2], fmt, mbs, blks)
    ctx = pkg.Context(w, h, yuv_format=fmt, max_refs=1, search_range=8)
    ctx.recon_upload(*planes)
    ctx.deblock_frame(mbs, blks)
    got = ctx.recon_download()
    ctx.close()
    changed = int((want[0] != planes[0]).sum())
    assert changed > 0, "the case does not exercise the filter"
    for g, wv, name in zip(got, want, "YUV"):
        if wv is None:
            continue
        bad = np.argwhere(g != wv)
        assert bad.size == 0, "%s differs at %d samples, first (row, col) %s" % (name, len(bad), bad[0])
    return changed


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [1, 2, 3, 0])
@pytest.mark.parametrize("w,h", [(64, 48), (176, 144)])
def test_deblock_matches_oracle(pkg, fmt, w, h):
    run(pkg, w, h, fmt, seed=fmt * 10 + w)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(idc_mode="zero", intra_frac=0.0, t8_frac=0.0), dict(idc_mode="two", intra_frac=0.5),
                                dict(idc_mode="zero", intra_frac=1.0, qp_lo=40, qp_hi=51), dict(bslice=True, intra_frac=0.05),
                                dict(smooth=False, qp_lo=44, qp_hi=51, idc_mode="zero"), dict(idc_mode="zero", qp_lo=0, qp_hi=18)])
def test_deblock_variants(pkg, kw):
    if kw.get("qp_hi") == 18:
        # alpha = beta = 0 for indexA/B < 16: nothing may change (checked by the "exercise" assert being inverted)
        rng = np.random.default_rng(3)
        planes, mbs, blks = make_case(pkg, rng, 64, 64, 1, **kw)
        mbs["alpha_c0_offset"] = mbs["beta_offset"] = -4
        ctx = pkg.Context(64, 64, yuv_format=1, max_refs=1, search_range=8)
        ctx.recon_upload(*planes)
        ctx.deblock_frame(mbs, blks)
        got = ctx.recon_download()
        ctx.close()
        for g, p in zip(got, planes):
            assert np.array_equal(g, p)
        return
    run(pkg, 128, 96, 1, seed=11, **kw)


@pytest.mark.gpu
def test_deblock_1080p_and_wide_diagonals(pkg, monkeypatch):
    """1080p (120 macroblocks per row, 60 per diagonal) and a 4K-wide strip (240 per row: more than 64 macroblocks on a diagonal,
    the kernel's multi-pass case)."""
    run(pkg, 1920, 1088, 1, seed=5)
    run(pkg, 3840, 1088, 1, seed=6, idc_mode="zero")
    # 4:2:2 with more than 64 macroblocks on a diagonal (136 x 66 macroblocks: 66): LDS ring in two passes, then the global-memory kernel
    run(pkg, 2176, 1056, 2, seed=7, idc_mode="zero")
    monkeypatch.setenv("JMHIP_DEBLOCK_KERNEL", "global")
    run(pkg, 2176, 1056, 2, seed=7, idc_mode="zero")


@pytest.mark.gpu
def test_deblock_band_of_a_slice(pkg):
    """A band of rows on its own equals the same rows of the whole-picture result when the band is a slice with idc 2."""
    rng = np.random.default_rng(8)
    w, h = 160, 144
    planes, mbs, blks = make_case(pkg, rng, w, h, 1, idc_mode="two")
    want = oracle.deblock_frame(planes[0], planes[1], planes[2], 1, mbs, blks)
    ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=8)
    ctx.recon_upload(*planes)
    sr = max(1, (h // 16) // 3)
    for r0 in range(0, h // 16, sr):                      # slice by slice, in any order: they do not touch each other
        ctx.deblock_frame(mbs, blks, 4, r0, min(sr, h // 16 - r0))
    got = ctx.recon_download()
    ctx.close()
    for g, wv in zip(got, want):
        assert np.array_equal(g, wv)


@pytest.mark.gpu
def test_deblock_bands_in_sequence_cross_their_borders(pkg):
    """Bands filtered one after the other (top to bottom) give the whole-picture result even when the filter crosses the band
    borders (idc 0): the first row of a band then changes rows that only live in HBM."""
    rng = np.random.default_rng(9)
    w, h = 192, 160
    planes, mbs, blks = make_case(pkg, rng, w, h, 1, idc_mode="zero")
    want = oracle.deblock_frame(planes[0], planes[1], planes[2], 1, mbs, blks)
    ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=8)
    ctx.recon_upload(*planes)
    for r0, n in ((0, 3), (3, 1), (4, 6)):
        ctx.deblock_frame(mbs, blks, 4, r0, n)
    got = ctx.recon_download()
    ctx.close()
    for g, wv in zip(got, want):
        assert np.array_equal(g, wv)


@pytest.mark.gpu
@pytest.mark.parametrize("sched,sweeps", [("wave", None), ("relax", "4"), ("relax", None)])
def test_deblock_schedules_agree(pkg, sched, sweeps, monkeypatch):
    """4:2:0 pictures are filtered by RELAXATION by default (deblock.hip deblock_relax_kernel: every macroblock from what its left / up /
    up-right neighbours last produced, sweep after sweep until nothing changes, then one compose pass); JMHIP_DEBLOCK_SCHED=wave forces the
    2:1 wavefront kernels, JMHIP_DEBLOCK_SWEEPS caps the sweeps (4 = one group: pictures that need more fall through to the wavefront with
    the picture untouched). All three against the oracle: worst case (every edge filtered, strong intra edges), slices, 1080p."""
    monkeypatch.setenv("JMHIP_DEBLOCK_SCHED", sched)
    if sweeps:
        monkeypatch.setenv("JMHIP_DEBLOCK_SWEEPS", sweeps)
    run(pkg, 176, 144, 1, seed=41, idc_mode="zero", intra_frac=1.0, qp_lo=40, qp_hi=51)
    run(pkg, 320, 192, 1, seed=42, idc_mode="two", intra_frac=0.3)
    run(pkg, 1920, 1088, 1, seed=43, smooth=False, qp_lo=44, qp_hi=51, idc_mode="zero")
    run(pkg, 176, 144, 2, seed=44)
    run(pkg, 320, 64, 0, seed=45)


@pytest.mark.gpu
def test_deblock_global_memory_kernel_on_420(pkg, monkeypatch):
    """4:0:0 / 4:2:0 / 4:2:2 normally take the LDS-ring kernel; the global-memory wavefront kernel (4:4:4 / oversized pictures) must agree."""
    monkeypatch.setenv("JMHIP_DEBLOCK_KERNEL", "global")
    monkeypatch.setenv("JMHIP_DEBLOCK_SCHED", "wave")
    run(pkg, 176, 144, 1, seed=21)
    run(pkg, 320, 64, 0, seed=22)
    run(pkg, 176, 144, 2, seed=23)
    run(pkg, 176, 144, 3, seed=24, idc_mode="zero")


@pytest.mark.gpu
@pytest.mark.parametrize("idc,slice_rows", [(0, 0), (2, 2)])
def test_deblock_recon_from_the_frame_stage(pkg, idc, slice_rows):
    recon_stage_case(pkg, idc, slice_rows, 96, 64, 8)


@pytest.mark.gpu
def test_deblock_recon_2160p_eight_slices(pkg):
    """BASELINE config 4's layout at size: 3840x2160 cut into 8 slices of whole macroblock rows (17 rows each, the last 16), filter kept inside
    the slices (idc 2): search -> residual -> filter on the device, against the oracle's DeblockFrame on the same side information."""
    recon_stage_case(pkg, 2, 17, 3840, 2160, 16, spread=10)


def recon_stage_case(pkg, idc, slice_rows, w, h, R, spread=6):
    """jmhip_deblock_recon builds the filter's side information on the device from the results of me_frame + residual_frame.
    Expected: the oracle's filter on the downloaded (unfiltered) reconstruction with the same information assembled on the host."""
    from tests.test_frame import synth
    from tests.test_me import lambda_factors, make_mbs
    rng = np.random.default_rng(12)
    qp = 38
    cur, ref = synth(rng, w, h, 1)
    ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    ctx.ref_upload(0, *ref)
    ctx.interp_luma(0)
    ctx.interp_chroma(0)
    ctx.cur_upload(*cur)
    mbw, mbh = w // 16, h // 16
    mbs_me = make_mbs(pkg, rng, mbw, mbh, spread, per_partition=(w < 1000))
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lambda_factors(qp)
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    me = ctx.me_frame(prm, mbs_me)
    quants = np.array([pkg.flat_quant(qp + d, 342, adaptive_rounding=0, adapt_rnd_weight=4, cavlc=1) for d in (0, 0, 3)], dtype=pkg.QUANT_DTYPE)
    ctx.residual_frame(quants, None)
    got = ctx.residual_download(len(mbs_me), want_results=(w < 1000))
    before = ctx.recon_download()
    # the same information, assembled on the host
    mbs = np.zeros(mbw * mbh, pkg.DEBLOCK_MB_DTYPE)
    blks = np.zeros(16 * mbw * mbh, pkg.DEBLOCK_BLK_DTYPE)
    blks["ref_id"][:, 1] = INT64_MIN
    for i, job in enumerate(mbs_me):
        x, y = int(job["mb_x"]), int(job["mb_y"])
        a = y * mbw + x
        mbs["qp"][a], mbs["qpc"][a] = qp, (qp - 2, qp - 3)
        mbs["disable_idc"][a], mbs["alpha_c0_offset"][a], mbs["beta_offset"][a] = idc, 2, -2
        mbs["transform_8x8"][a] = int(got["modes"][i]["pad"][0]) != 0
        mbs["avail_a"][a] = x != 0
        mbs["avail_b"][a] = y != 0 and (slice_rows == 0 or y % slice_rows != 0)
        mbs["cbp_blk"][a] = int(got["cbp_blk"][i]) & 0xffff
        for t in range(16):
            p = oracle.covering_partition(got["modes"][i], t & 3, t >> 2)
            k = (y * 4 + (t >> 2)) * (mbw * 4) + x * 4 + (t & 3)
            blks["mv"][k, 0] = me["mv"][i, p]
            blks["ref_id"][k, 0] = int(job["ref"])
    want = oracle.deblock_frame(before[0], before[1], before[2], 1, mbs, blks)
    assert int((want[0] != before[0]).sum()) > 0
    ctx.deblock_recon(qp, (qp - 2, qp - 3), disable_idc=idc, alpha_c0_offset=2, beta_offset=-2, slice_rows=slice_rows)
    after = ctx.recon_download()
    ctx.close()
    for g, wv, name in zip(after, want, "YUV"):
        assert np.array_equal(g, wv), name


@pytest.mark.gpu
def test_deblock_one_workgroup_per_slice_equals_one_workgroup(pkg, monkeypatch):
    """Slices whose first row filters no top edge (idc 2, or idc 1) are independent bands: the library gives each its own workgroup.
    Same result as the single-workgroup walk (JMHIP_DEBLOCK_BANDS=0) and as the oracle, 1080p with 8 slices and a mixed-idc picture."""
    for kw in (dict(idc_mode="two", nslices=8), dict(idc_mode="mixed", nslices=5)):
        run(pkg, 1920, 1088, 1, seed=31, **kw)
        monkeypatch.setenv("JMHIP_DEBLOCK_BANDS", "0")
        run(pkg, 1920, 1088, 1, seed=31, **kw)
        monkeypatch.delenv("JMHIP_DEBLOCK_BANDS")
    run(pkg, 320, 1088, 2, seed=32, idc_mode="two", nslices=68)     # one slice per macroblock row: more bands than workgroups (64)
