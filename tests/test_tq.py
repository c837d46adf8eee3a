"""dct_4x4 / dct_8x8 / dct_16x16 / dct_chroma: jmhip_tq_batch vs the oracle, bit-exact on levels, runs (up to the
terminating zero level), reconstruction, coefficient cost, nonzero flags, adaptive-rounding residues, return values."""
import numpy as np
import pytest

from tests import oracle


def make_jobs(pkg, rng, n, nq, amp, chroma=False):
    jobs = np.zeros(n, dtype=pkg.TQ_JOB_DTYPE)
    pred = rng.integers(0, 256, (n, 16, 16))
    resid = np.round(rng.normal(0, amp, (n, 16, 16))).astype(int)
    # a few blocks with zero residual, a few with saturating recon
    resid[::7] = 0
    jobs["pred"] = pred
    jobs["src"] = np.clip(pred + resid, 0, 255)
    jobs["quant"] = rng.integers(0, nq, n)
    jobs["quant_dc"] = jobs["quant"]
    if chroma:
        jobs["uv"] = rng.integers(0, 2, n)
        jobs["cr_cbp_in"] = rng.integers(0, 3, n)
    return jobs


def compare_lists(got_lev, got_run, want_lev, want_run, what):
    """(level, run) lists are compared up to and including the terminating zero level."""
    g, w = got_lev.reshape(-1, got_lev.shape[-1]), want_lev.reshape(-1, want_lev.shape[-1])
    gr, wr = got_run.reshape(g.shape), want_run.reshape(w.shape)
    for r in range(g.shape[0]):
        nz = np.flatnonzero(w[r] == 0)
        k = nz[0] if len(nz) else w.shape[1] - 1
        assert np.array_equal(g[r, :k + 1], w[r, :k + 1]), "%s levels row %d: %s vs %s" % (what, r, g[r, :k + 2], w[r, :k + 2])
        assert np.array_equal(gr[r, :k], wr[r, :k]), "%s runs row %d" % (what, r)


def quant_set(pkg, qps, offset11, is8x8=False, **kw):
    return np.array([pkg.flat_quant(qp, offset11, is8x8, **kw) for qp in qps], dtype=pkg.QUANT_DTYPE)


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(64, 48, yuv_format=1)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("field_scan,adaptive,amp", [(0, 1, 6), (1, 0, 20), (0, 0, 60), (0, 1, 1)])
def test_dct_4x4(pkg, ctx, field_scan, adaptive, amp):
    rng = np.random.default_rng(amp + field_scan)
    quants = quant_set(pkg, [4, 17, 28, 36, 51], 342, adaptive_rounding=adaptive, adapt_rnd_weight=4, field_scan=field_scan, cavlc=1)
    jobs = make_jobs(pkg, rng, 300, len(quants), amp)
    got = ctx.tq_batch("luma4x4", quants, jobs)
    want = oracle.tq_reference("luma4x4", quants, jobs)
    compare_lists(got["levels"], got["runs"], want["levels"], want["runs"], "dct_4x4")
    for k in ("recon", "coeff_cost", "nonzero"):
        assert np.array_equal(got[k], want[k]), k
    if adaptive:
        assert np.array_equal(got["fadjust"], want["fadjust"])


@pytest.mark.gpu
@pytest.mark.parametrize("cavlc,flag,field_scan,adaptive", [(1, 1, 0, 1), (0, 1, 0, 0), (1, 0, 1, 1), (0, 1, 1, 1)])
def test_dct_8x8(pkg, ctx, cavlc, flag, field_scan, adaptive):
    rng = np.random.default_rng(cavlc * 8 + flag * 4 + field_scan)
    quants = quant_set(pkg, [6, 20, 28, 40], 342, True, adaptive_rounding=adaptive, adapt_rnd_weight=4, field_scan=field_scan,
                       cavlc=cavlc, transform8x8_flag=flag)
    jobs = make_jobs(pkg, rng, 200, len(quants), 15)
    got = ctx.tq_batch("luma8x8", quants, jobs)
    want = oracle.tq_reference("luma8x8", quants, jobs)
    if cavlc and flag:
        compare_lists(got["levels"], got["runs"], want["levels"], want["runs"], "dct_8x8 interleaved")
    else:
        compare_lists(got["levels8"], got["runs8"], want["levels8"], want["runs8"], "dct_8x8")
    for k in ("recon",):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["coeff_cost"][:, :4], want["coeff_cost"][:, :4])
    assert np.array_equal(got["nonzero"][:, :4], want["nonzero"][:, :4])
    if adaptive:
        assert np.array_equal(got["fadjust"], want["fadjust"])


@pytest.mark.gpu
@pytest.mark.parametrize("qps,cavlc,amp", [([5, 8, 28, 44], 1, 25), ([20, 30], 0, 8)])
def test_dct_16x16(pkg, ctx, qps, cavlc, amp):
    rng = np.random.default_rng(sum(qps))
    quants = quant_set(pkg, qps, 682, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=cavlc)
    jobs = make_jobs(pkg, rng, 150, len(quants), amp)
    got = ctx.tq_batch("luma16x16", quants, jobs)
    want = oracle.tq_reference("luma16x16", quants, jobs)
    compare_lists(got["dc_levels"][:, None, :], got["dc_runs"][:, None, :], want["dc_levels"][:, None, :], want["dc_runs"][:, None, :], "dct_16x16 DC")
    compare_lists(got["levels"][:, :, :16], got["runs"][:, :, :16], want["levels"][:, :, :16], want["runs"][:, :, :16], "dct_16x16 AC")
    for k in ("recon", "ret", "fadjust"):
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,amp", [(1, 10), (1, 2), (2, 10), (2, 2)])
def test_dct_chroma(pkg, ctx, fmt, amp):
    rng = np.random.default_rng(fmt * 100 + amp)
    qps = [3, 12, 24, 33, 39]
    quants = np.concatenate([quant_set(pkg, qps, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1),
                             quant_set(pkg, [q + 3 for q in qps], 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1)])
    quants["img_qp"] = np.concatenate([qps, qps])
    jobs = make_jobs(pkg, rng, 300, len(qps), amp, chroma=True)
    jobs["quant_dc"] = jobs["quant"] + len(qps)      # the 4:2:2 DC quantiser is the qp+3 table set
    got = ctx.tq_batch("chroma", quants, jobs, yuv_format=fmt)
    want = oracle.tq_reference("chroma", quants, jobs, yuv_format=fmt)
    nblk = 4 if fmt == 1 else 8
    rows, cols = (8, 8) if fmt == 1 else (16, 8)
    compare_lists(got["dc_levels"][:, None, :], got["dc_runs"][:, None, :], want["dc_levels"][:, None, :], want["dc_runs"][:, None, :], "dct_chroma DC")
    compare_lists(got["levels"][:, :nblk, :16], got["runs"][:, :nblk, :16], want["levels"][:, :nblk, :16], want["runs"][:, :nblk, :16], "dct_chroma AC")
    assert np.array_equal(got["recon"][:, :rows, :cols], want["recon"][:, :rows, :cols])
    assert np.array_equal(got["ret"], want["ret"])
    assert np.array_equal(got["fadjust"][:, :rows, :cols], want["fadjust"][:, :rows, :cols])
    assert np.array_equal(got["cbp_blk"], want["cbp_blk"] & ~want["cbp_clear"])
    assert np.array_equal(got["cbp_clear"], want["cbp_clear"])


@pytest.mark.gpu
def test_flat_tables_match_oracle(pkg):
    for qp in (0, 7, 28, 51):
        for is8 in (False, True):
            q = pkg.flat_quant(qp, 342, is8)
            ls, ils, lo, n = oracle.flat_tables(qp, 342, is8)
            assert np.array_equal(q["levelscale"][:n], ls[:n]) and np.array_equal(q["invlevelscale"][:n], ils[:n])
            assert np.array_equal(q["leveloffset"][:n], lo[:n])
