"""The HIP path (through the C ABI) against golden vectors captured from the REAL JM (tests/golden/*.npz)."""
import numpy as np
import pytest

from tests import oracle
from tests.test_golden import GOLD, IDS, fnv, lists_equal, luma_pic, quant_from, recs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_interp_planes_match_jm(pkg, path):
    z = np.load(path)
    luma, chroma = recs(z, "luma"), recs(z, "chroma")
    for lr, cr in zip(luma, chroma):
        Y, w, h = luma_pic(lr)
        wc, hc, fmt = int(cr[1]), int(cr[2]), int(cr[3])
        (sx, sy), _ = oracle.chroma_geom(fmt)
        U = cr[4:4 + wc * hc].reshape(hc, wc)
        off = 4 + wc * hc + sx * sy
        V = cr[off:off + wc * hc].reshape(hc, wc)
        ctx = pkg.Context(w, h, yuv_format=fmt)
        ctx.ref_upload(0, Y.astype(np.uint16), U.astype(np.uint16), V.astype(np.uint16))     # JM's imgpel is 16-bit
        ctx.interp_luma(0)
        ctx.interp_chroma(0)
        planes = ctx.download_luma_planes(0, dtype=np.uint16)
        dig = lr[4 + w * h:4 + w * h + 16].astype(np.uint32)
        assert np.array_equal(np.array([fnv(planes[p >> 2, p & 3]) for p in range(16)], dtype=np.uint32), dig)
        for uv, o in ((0, 4 + wc * hc), (1, off + wc * hc)):
            cp = ctx.download_chroma_planes(0, uv, dtype=np.uint16)
            want = cr[o:o + sx * sy].astype(np.uint32)
            assert np.array_equal(np.array([fnv(cp[y, x]) for y in range(sy) for x in range(sx)], dtype=np.uint32), want)
        ctx.close()


def _search_ctx(pkg, z, r, R):
    lr = recs(z, "luma")[int(r[0])]
    Y, w, h = luma_pic(lr)
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=max(R, 1))
    ctx.ref_upload(0, Y.astype(np.uint8))
    return ctx, w, h


def _partition_of(pkg, bt, x4, y4):
    for i, (b, x, y, _, _) in enumerate(pkg.partition_table()):
        if (b, x, y) == (bt, x4, y4):
            return i
    raise AssertionError("no partition")


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_integer_search_matches_jm(pkg, path):
    """FullPelBlockMotionSearch and FastFullPelBlockMotionSearch records replayed through jmhip_me_frame (integer stage)."""
    z = np.load(path)
    done = 0
    for kind in ("fullpel", "fastfull"):
        for r in recs(z, kind):
            if int(r[3]) != 0:                                              # records with the chroma term carry no chroma planes
                continue
            a = r[14:]
            if kind == "fullpel":
                ref0, px, py, bt, pmx, pmy, ix, iy, R, minc, lam, ox, oy, ocost = [int(v) for v in a[:14]]
                blk = a[14:]
                mode, mbx, mby = -1, px >> 4, py >> 4
                cpx, cpy = pmx, pmy
            else:
                ref0, ox0, oy0, px, py, bt, cpx, cpy, pmx, pmy, R, minc, lam, ox, oy, ocost = [int(v) for v in a[:16]]
                blk = a[16:16 + 256]
                mode, mbx, mby = 0, ox0 >> 4, oy0 >> 4
            if minc != 2147483647:
                continue
            ctx, w, h = _search_ctx(pkg, z, r, R)
            # the current picture only matters inside the searched block
            cur = np.zeros((h, w), np.uint8)
            bsx, bsy = oracle.BLOCK_SIZE[bt]
            if kind == "fullpel":
                cur[py:py + bsy, px:px + bsx] = blk.reshape(bsy, bsx)
            else:
                cur[oy0:oy0 + 16, ox0:ox0 + 16] = blk.reshape(16, 16)
            ctx.cur_upload(cur)
            p = _partition_of(pkg, bt, (px & 15) >> 2, (py & 15) >> 2)
            mbs = np.zeros(1, dtype=pkg.ME_MB_DTYPE)
            mbs[0]["mb_x"], mbs[0]["mb_y"], mbs[0]["ref"], mbs[0]["ref_is_0"] = mbx, mby, 0, ref0
            mbs[0]["pred_mv"][:] = (cpx, cpy)              # FastFull: partition 0 carries the 16x16 predictor of the window centre
            mbs[0]["pred_mv"][p] = (pmx, pmy)
            prm = pkg.MeParams()
            prm.search_mode, prm.search_range, prm.rdopt, prm.is_b_slice = mode, R, int(r[1]), int(r[2])
            prm.level_mv_min, prm.level_mv_max = -511, 511
            prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = lam
            prm.transform8x8_mode, prm.subpel = int(r[4]), 0
            prm.partition_mask = (1 << p) | (1 if mode == 0 else 0)
            prm.metric_set = 1                              # MEErrorMetric[] of the record (SAD / SSE / Hadamard SAD at integer positions)
            prm.metric[0], prm.metric[1], prm.metric[2] = int(r[5]), int(r[6]), int(r[7])
            if int(r[8]):                                   # UseWeightedReferenceME: weight_luma, offset_luma, wp_luma_round, luma_log_weight_denom
                prm.wp_enable, prm.wp_round, prm.wp_denom = 1, int(r[11]), int(r[12])
                prm.wp_weight[0], prm.wp_offset[0] = int(r[9]), int(r[10])
            got = ctx.me_frame(prm, mbs)[0]
            ctx.close()
            assert (int(got["mv_int"][p][0]), int(got["mv_int"][p][1]), int(got["cost_int"][p])) == (ox, oy, ocost), (kind, px, py, bt)
            done += 1
    assert done > 0 or "fullpel" not in z


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_subpel_search_matches_jm(pkg, path):
    """SubPelBlockMotionSearch records of the real JM replayed through jmhip_me_subpel: every recorded metric triple, the carried minimum
    (min_mcost of the record) when the integer and half-pel metric agree."""
    z = np.load(path)
    done = 0
    for r in recs(z, "subpel"):
        if int(r[3]) != 0:
            continue
        a = r[14:]
        ref0, px, py, bt, pmx, pmy, ix, iy, sp2, sp4, minc, l0, l1, l2, ox, oy, ocost = [int(v) for v in a[:17]]
        start_hp = int(r[5]) == int(r[6])
        if sp2 != 9 or sp4 != 9 or (ix | iy) & 3 or (minc != 2147483647 and not start_hp):
            continue
        lr = recs(z, "luma")[int(r[0])]
        Y, w, h = luma_pic(lr)
        ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=16)
        ctx.ref_upload(0, Y.astype(np.uint8))
        ctx.interp_luma(0)
        bsx, bsy = oracle.BLOCK_SIZE[bt]
        cur = np.zeros((h, w), np.uint8)
        cur[py:py + bsy, px:px + bsx] = a[17:17 + bsx * bsy].reshape(bsy, bsx)
        ctx.cur_upload(cur)
        p = _partition_of(pkg, bt, (px & 15) >> 2, (py & 15) >> 2)
        mbs = np.zeros(1, dtype=pkg.ME_MB_DTYPE)
        mbs[0]["mb_x"], mbs[0]["mb_y"], mbs[0]["ref"], mbs[0]["ref_is_0"] = px >> 4, py >> 4, 0, ref0
        mbs[0]["pred_mv"][p] = (pmx, pmy)
        prm = pkg.MeParams()
        prm.search_mode, prm.search_range, prm.rdopt, prm.is_b_slice = -1, 16, int(r[1]), int(r[2])
        prm.level_mv_min, prm.level_mv_max = -511, 511
        prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = l0, l1, l2
        prm.transform8x8_mode, prm.subpel, prm.partition_mask = int(r[4]), 1, 1 << p
        prm.metric_set = 1
        prm.metric[0], prm.metric[1], prm.metric[2] = int(r[5]), int(r[6]), int(r[7])
        if int(r[8]):
            prm.wp_enable, prm.wp_round, prm.wp_denom = 1, int(r[11]), int(r[12])
            prm.wp_weight[0], prm.wp_offset[0] = int(r[9]), int(r[10])
        res = np.zeros(1, dtype=pkg.ME_RESULT_DTYPE)
        res[0]["mv_int"][p] = (ix >> 2, iy >> 2)
        res[0]["cost_int"][p] = minc
        got = ctx.me_subpel(prm, mbs, res)[0]
        ctx.close()
        assert (int(got["mv"][p][0]), int(got["mv"][p][1]), int(got["cost"][p])) == (ox, oy, ocost), (px, py, bt, tuple(int(v) for v in r[5:8]))
        done += 1
    assert done > 0 or len(recs(z, "subpel")) == 0


def _tile_job(pkg, m7, mpr):
    job = np.zeros(1, dtype=pkg.TQ_JOB_DTYPE)
    job[0]["pred"] = mpr
    job[0]["src"] = np.clip(m7 + mpr, 0, 255)           # exact inside the coded block (src = original sample)
    return job


def _q(pkg, q):
    out = np.zeros(1, dtype=pkg.QUANT_DTYPE)
    for name in out.dtype.names:
        out[0][name] = q[name]
    return out


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_transform_quant_matches_jm(pkg, path):
    z = np.load(path)
    ctx = pkg.Context(64, 48, yuv_format=1)
    for r in recs(z, "dct4"):
        bx, by, intra, cc_in = [int(v) for v in r[:4]]
        q, a = quant_from(r[4:], 4)
        m7, mpr, o = a[:256].reshape(16, 16), a[256:512].reshape(16, 16), a[512:]
        src = m7 + mpr
        if src[by:by + 4, bx:bx + 4].min() < 0 or src[by:by + 4, bx:bx + 4].max() > 255:
            continue
        g = ctx.tq_batch("luma4x4", _q(pkg, q), _tile_job(pkg, m7, mpr))[0]
        blk = (2 * (by >> 3) + (bx >> 3)) * 4 + 2 * ((by >> 2) & 1) + ((bx >> 2) & 1)
        assert int(g["nonzero"][blk]) == int(o[0]) and cc_in + int(g["coeff_cost"][blk]) == int(o[1])
        assert lists_equal(g["levels"][blk], g["runs"][blk], o[2:19], o[19:36])
        assert np.array_equal(g["recon"][by:by + 4, bx:bx + 4], o[36:52].reshape(4, 4))
    for r in recs(z, "dct8"):
        b8, intra, cc_in = [int(v) for v in r[:3]]
        q, a = quant_from(r[3:], 8)
        m7, mpr, o = a[:256].reshape(16, 16), a[256:512].reshape(16, 16), a[512:]
        ys, xs = 8 * (b8 >> 1), 8 * (b8 & 1)
        g = ctx.tq_batch("luma8x8", _q(pkg, q), _tile_job(pkg, m7, mpr))[0]
        assert int(g["nonzero"][b8]) == int(o[0]) and cc_in + int(g["coeff_cost"][b8]) == int(o[1])
        lr = o[2:2 + 520].reshape(4, 2, 65)
        if q["transform8x8_flag"] and q["cavlc"]:
            for k in range(4):
                assert lists_equal(g["levels"][4 * b8 + k], g["runs"][4 * b8 + k], lr[k, 0][:17], lr[k, 1][:17])
        else:
            assert lists_equal(g["levels8"][b8], g["runs8"][b8], lr[0, 0], lr[0, 1])
        assert np.array_equal(g["recon"][ys:ys + 8, xs:xs + 8], o[522:586].reshape(8, 8))
    for r in recs(z, "dct16"):
        q, a = quant_from(r[1:], 4)
        job = np.zeros(1, dtype=pkg.TQ_JOB_DTYPE)
        job[0]["src"], job[0]["pred"] = a[:256].reshape(16, 16), a[256:512].reshape(16, 16)
        o = a[512:]
        g = ctx.tq_batch("luma16x16", _q(pkg, q), job)[0]
        assert int(g["ret"]) == int(o[0])
        assert lists_equal(g["dc_levels"], g["dc_runs"], o[1:18], o[18:35])
        ac = o[35:35 + 512].reshape(16, 2, 16)
        for b in range(16):
            assert lists_equal(g["levels"][b][:16], g["runs"][b][:16], ac[b, 0], ac[b, 1])
        assert np.array_equal(g["recon"], o[547:547 + 256].reshape(16, 16))
    for r in recs(z, "dctc"):
        uv, cr_in, fmt = int(r[0]), int(r[1]), int(r[2])
        q, a = quant_from(r[5:], 4)
        qdc, a = quant_from(a, 4)
        m7, mpr, o = a[:256].reshape(16, 16), a[256:512].reshape(16, 16), a[512:]
        rows, cols = (8, 8) if fmt == 1 else (16, 8)
        src = m7 + mpr
        if src[:rows, :cols].min() < 0 or src[:rows, :cols].max() > 255:
            continue
        job = _tile_job(pkg, m7, mpr)
        job[0]["quant"], job[0]["quant_dc"], job[0]["uv"], job[0]["cr_cbp_in"] = 0, 1, uv, cr_in
        g = ctx.tq_batch("chroma", np.concatenate([_q(pkg, q), _q(pkg, qdc)]), job, yuv_format=fmt)[0]
        assert int(g["ret"]) == int(o[0])
        cbp_in = (int(r[3]) & 0xffffffff) | (int(r[4]) << 32)
        cbp_out = (int(o[1]) & 0xffffffff) | (int(o[2]) << 32)
        assert ((cbp_in & ~int(g["cbp_clear"])) | int(g["cbp_blk"])) & 0xffffffffffffffff == cbp_out & 0xffffffffffffffff
        assert lists_equal(g["dc_levels"], g["dc_runs"], o[3:20], o[20:37])
        ac = o[37:37 + 256].reshape(8, 2, 16)
        for b in range(4 if fmt == 1 else 8):
            assert lists_equal(g["levels"][b][:16], g["runs"][b][:16], ac[b, 0], ac[b, 1])
        assert np.array_equal(g["recon"][:rows, :cols], o[293:293 + rows * cols].reshape(rows, cols))
    ctx.close()
