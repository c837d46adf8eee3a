"""jmhip_p_slice_search (me_wave.hip): a whole P slice on the device -- predictors, FullSearch / FastFull / UMHexagonS / EPZS integer search,
sub-pel refinement, skip shortcut and the low-complexity inter decision as a macroblock wavefront -- bit-exact against
  (1) the REAL JM: the frame-level fixtures tests/golden/field_*.npz (every BlockMotionSearch call + the final field of each P picture),
  (2) the oracle's driver (oracle/jmo_lowcplx.c, itself pinned on those fixtures) on larger synthetic pictures, several references, slices."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle
from tests.test_golden_field import CASES, GOLD, QP_N, part_index

FIELDS = ("pred", "mv_int", "cost_int", "mv", "cost")


def slice_params(pkg, mode, R, nref, lambda_mf, ref_cost1, W, mb_first=0, mb_count=None, H=None, metric=(0, 2, 2), qp_n=QP_N, full_search=2):
    lib = pkg.load_library()
    p = pkg.SliceParams()
    p.search_mode, p.search_range, p.full_search, p.num_refs = mode, R, full_search, nref
    for r in range(nref):
        p.ref_slot[r] = r
    for m in range(1, 8):
        p.valid[m] = 1
    p.lambda_mf[0], p.lambda_mf[1], p.lambda_mf[2] = lambda_mf
    p.ref_cost1, p.md_metric = ref_cost1, 2
    p.metric[0], p.metric[1], p.metric[2] = metric
    p.level_mv_min, p.level_mv_max = -511, 511
    p.mb_first = mb_first
    p.mb_count = mb_count if mb_count is not None else (W // 16) * (H // 16) - mb_first
    lib.jmhip_epzs_setup(p, R, 2, 3, 2, 1, 1, 1, 0, 1, 2, 2)
    lib.jmhip_umhex_setup(p, 1, 3, qp_n, W)
    return p


def compare(got, want, nref, what):
    for f in ("best_mode", "min_cost", "b8mode", "b8ref", "final_mv", "skip_mv"):
        if not np.array_equal(got[f], want[f]):
            i = int(np.argwhere(np.any((got[f] != want[f]).reshape(len(got), -1), axis=1))[0][0])
            raise AssertionError("%s: %s differs at macroblock %d: device %s oracle %s" % (what, f, i, got[f][i].tolist(), want[f][i].tolist()))
    for f in FIELDS:
        g, w = got[f][:, :nref], want[f][:, :nref]
        if not np.array_equal(g, w):
            idx = np.argwhere((g != w).reshape(len(got), nref, 41, -1).any(axis=3))[0]
            raise AssertionError("%s: %s differs at macroblock %d ref %d partition %d: device %s oracle %s" % (what, f, idx[0], idx[1], idx[2], g[tuple(idx)].tolist(), w[tuple(idx)].tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_fixture_pictures_match_the_real_jm(pkg, name):
    mode = CASES[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    n = int(z["n_frames"])
    W, H = int(z["f0_head"][0]), int(z["f0_head"][1])
    R = 16
    ctx = pkg.Context(W, H, yuv_format=0, max_refs=2, search_range=R)
    ctx.slice_state_reset()
    epzs = oracle.Epzs(W, H, R, 2) if mode == 3 else None         # only its EPZSSliceInit: the co-located field is an INPUT of the device call
    lib = pkg.load_library()
    for k in range(n):
        head = z["f%d_head" % k]
        nref = int(head[3])
        refinfo = z["f%d_refinfo" % k]
        for r in range(nref):
            ctx.ref_upload(r, z["f%d_refs" % k][r])
            ctx.interp_luma(r)
        ctx.cur_upload(z["f%d_cur" % k])
        p = slice_params(pkg, mode, R, nref, [int(v) for v in head[6:9]], int(head[9]), W, H=H)
        if mode == 3:
            ids = (refinfo[:, 1].astype(np.int64) & 0xffffffff) | (refinfo[:, 2].astype(np.int64) << 32)
            epzs.slice_init(int(head[10]), [int(v) for v in refinfo[:, 0]], ids, z["f%d_col_mv" % k], z["f%d_col_ref_id" % k], num_ref_idx_l0_active=int(head[11]))
            ctx.epzs_colocated_upload(oracle.epzs_colocated(epzs, W, H))
            pocs = [int(v) for v in refinfo[:, 0]]
            lib.jmhip_epzs_scales(p, int(head[10]), (C.c_int * nref)(*pocs), nref)
        got = ctx.p_slice_search(p)
        bad = 0
        for (mb, ref, bt, bx, by, px, py, mx, my, cost, rng, lam) in z["f%d_calls" % k]:
            pi = part_index(int(bt), int(bx), int(by))
            g = got[int(mb)]
            t = (int(g["pred"][ref, pi, 0]), int(g["pred"][ref, pi, 1]), int(g["mv"][ref, pi, 0]), int(g["mv"][ref, pi, 1]), int(g["cost"][ref, pi]))
            if t != (px, py, mx, my, cost):
                if not bad:
                    first = "picture %d mb %d ref %d blocktype %d block (%d,%d): JM pred (%d,%d) mv (%d,%d) cost %d, device %s" % (k, mb, ref, bt, bx, by, px, py, mx, my, cost, t)
                bad += 1
        assert bad == 0, "%d of %d BlockMotionSearch calls differ from JM; first: %s" % (bad, len(z["f%d_calls" % k]), first)
        ref_idx, mv = ctx.slice_field()
        fld = z["f%d_field" % k]
        assert np.array_equal(ref_idx, fld[..., 0]) and np.array_equal(mv, fld[..., 1:]), "picture %d: final field differs from JM" % k
    if epzs:
        epzs.close()
    ctx.close()


def synth_clip(rng, W, H, nframes):
    yy, xx = np.mgrid[0:H + 64, 0:W + 64]
    base = (np.sin(xx / 7.0) * np.cos(yy / 11.0) * 60 + np.sin((xx + 2 * yy) / 23.0) * 40 + 128 + rng.normal(0, 10, (H + 64, W + 64)))
    out = []
    for f in range(nframes):
        dx, dy = 3 * f, -2 * f
        fr = base[32 + dy:32 + dy + H, 32 + dx:32 + dx + W] + rng.normal(0, 2.5, (H, W))
        fr[H // 3:H // 2, W // 4:W // 2] = base[32 + H // 3 - 3 * f:32 + H // 2 - 3 * f, 32 + W // 4 + f:32 + W // 2 + f]     # a patch moving differently
        out.append(np.clip(np.round(fr), 0, 255).astype(np.uint8))
    return out


def run_synthetic(pkg, mode, W, H, R, nref, slices=1, nframes=3, seed=5, metric=(0, 2, 2), qp=28, max_mbs=None):
    """Frames 1.. are coded as P pictures against the previous `nref` SOURCE frames (the search does not care where a reference came from)."""
    rng = np.random.default_rng(seed)
    clip = synth_clip(rng, W, H, nframes + nref - 1)
    lam = int(65536 * np.sqrt(0.85 * 2 ** ((qp - 12) / 3.0) * (4 if False else 1)) + 0.5)     # any positive factor does
    lam3 = [lam, lam, lam]
    ref_cost1 = int(2 * np.sqrt(0.85 * 2 ** ((qp - 12) / 3.0)))
    nmb = (W // 16) * (H // 16)
    ctx = pkg.Context(W, H, yuv_format=0, max_refs=nref, search_range=R)
    ctx.slice_state_reset()
    epzs = oracle.Epzs(W, H, R, nref) if mode == 3 else None
    umhex = oracle.Umhex(W, H, R, nref, qp) if mode == 1 else None
    all_mv_state = np.zeros((4, 4, oracle.MAX_REFS, 9, 2), np.int16)
    prev_field = [(np.zeros((H // 4, W // 4, 2), np.int16), np.full((H // 4, W // 4), -1, np.int64))] * 2
    lib = pkg.load_library()
    passes = []
    for f in range(nref, nref + nframes - 1):
        cur = clip[f]
        refs = [clip[f - 1 - r] for r in range(nref)]
        pocs = [2 * (f - 1 - r) for r in range(nref)]
        orefs = [oracle.RefPic(r, yuv_format=0) for r in refs]
        for r in range(nref):
            ctx.ref_upload(r, refs[r])
            ctx.interp_luma(r)
        ctx.cur_upload(cur)
        if epzs:
            epzs.slice_init(2 * f, pocs, pocs, [prev_field[0][0], prev_field[1][0]], [prev_field[0][1], prev_field[1][1]])
            ctx.epzs_colocated_upload(oracle.epzs_colocated(epzs, W, H))
        ref_idx = np.full((H // 4, W // 4), -1, np.int8)
        mvf = np.zeros((H // 4, W // 4, 2), np.int16)
        per = (nmb + slices - 1) // slices
        for s in range(slices):
            first, count = s * per, min(per, nmb - s * per)
            q = oracle.lowcplx_params(mode, R, nref, lam3, ref_cost1, W, H, epzs=epzs, umhex=umhex, all_mv_state=all_mv_state, metric=metric)
            sid = (np.arange(nmb) // per).astype(np.int32)
            q._sid = sid
            q.slice_id = sid.ctypes.data
            want, _, _ = oracle.lowcplx_p_slice(q, orefs, cur, ref_idx, mvf, mb_first=first, mb_count=count)
            p = slice_params(pkg, mode, R, nref, lam3, ref_cost1, W, mb_first=first, mb_count=count, metric=metric, qp_n=qp)
            if mode == 3:
                lib.jmhip_epzs_scales(p, 2 * f, (C.c_int * nref)(*pocs), nref)
            got = ctx.p_slice_search(p)
            passes.append(ctx.slice_passes())
            compare(got, want, nref, "frame %d slice %d" % (f, s))
        gref, gmv = ctx.slice_field()
        assert np.array_equal(gref, ref_idx) and np.array_equal(gmv, mvf), "frame %d: final field" % f
        # what EPZS will read of this picture when it is the co-located one: its vectors and, as reference ids, the POCs of what they point to
        ids = np.where(ref_idx >= 0, np.array(pocs, np.int64)[np.clip(ref_idx, 0, None)], -1)
        prev_field = [(mvf.copy(), ids), prev_field[0]]
    if epzs:
        epzs.close()
    if umhex:
        umhex.close()
    ctx.close()
    return passes


@pytest.mark.gpu
@pytest.mark.parametrize("mode,W,H,R,nref,slices", [
    (-1, 96, 64, 8, 1, 1),          # FullSearch: per-partition centres from the true predictors
    (-1, 96, 64, 8, 2, 2),
    (0, 96, 64, 8, 2, 1),           # FastFullSearch
    (3, 176, 144, 16, 2, 1),        # EPZS, two references
    (3, 320, 192, 32, 3, 3),        # EPZS, +-32, three references, three slices (the last two begin mid-row)
    (1, 176, 144, 16, 2, 1),        # UMHexagonS with the dynamic search range
    (1, 320, 192, 32, 3, 3),
])
def test_synthetic_clip_matches_the_oracle(pkg, mode, W, H, R, nref, slices):
    run_synthetic(pkg, mode, W, H, R, nref, slices=slices)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [3, 1])
def test_sad_everywhere_and_satd_everywhere(pkg, mode):
    run_synthetic(pkg, mode, 176, 144, 16, 2, metric=(0, 0, 0), nframes=2)
    run_synthetic(pkg, mode, 176, 144, 16, 2, metric=(2, 2, 2), nframes=2)     # SATD at full-pel positions too (BASELINE config 3: "EPZS + SATD cost")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [3, 1])
def test_1080p_walkers_match_the_oracle(pkg, mode):
    """BASELINE configs 3 / 5 at their picture width: one 1920x1088 P picture, +-32, two references, every macroblock against the oracle."""
    passes = run_synthetic(pkg, mode, 1920, 1088, 32, 2, nframes=2, seed=11)
    print("passes per slice call:", passes)
