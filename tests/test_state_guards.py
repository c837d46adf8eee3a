"""State guards of the C ABI found by the round-1 code review: a failed job upload must not leave a resident job set behind,
and the recon picture is invalid after jmhip_recon_to_ref traded its planes with a reference slot."""
import numpy as np
import pytest

from tests.test_me import lambda_factors, make_mbs, make_pair


def _params(pkg, R):
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = -1, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lambda_factors(28)
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    return prm


@pytest.mark.gpu
def test_failed_upload_invalidates_resident_jobs(pkg):
    rng = np.random.default_rng(3)
    w, h, R = 64, 48, 8
    cur, ref = make_pair(rng, w, h, "shift")
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1, search_range=R)
    ctx.ref_upload(0, ref)
    ctx.interp_luma(0)
    ctx.cur_upload(cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, 6)
    prm = _params(pkg, R)
    first = ctx.me_frame(prm, mbs)
    ctx.me_frame_async(prm, None, len(mbs))                 # resident re-run is legal now
    again = ctx.me_results(len(mbs))
    bad = mbs.copy()
    bad[len(bad) // 2]["mb_x"] = 99                          # outside the picture: validation fails half-way through the list
    with pytest.raises(pkg.JmhipError):
        ctx.me_frame(prm, bad)
    with pytest.raises(pkg.JmhipError):                      # ... and the previous upload is no longer "resident"
        ctx.me_frame_async(prm, None, len(mbs))
    got = ctx.me_frame(prm, mbs)                             # a fresh upload works and gives the same answer
    for k in ("mv", "cost"):
        assert np.array_equal(got[k], first[k])
        assert np.array_equal(again[k], first[k])
    ctx.close()


@pytest.mark.gpu
def test_recon_invalid_after_plane_swap(pkg):
    rng = np.random.default_rng(4)
    w, h, R = 64, 48, 8
    from tests.test_frame import synth
    cur, ref = synth(rng, w, h, 1)
    ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    ctx.ref_upload(0, *ref)
    ctx.interp_luma(0)
    ctx.cur_upload(*cur)
    mbs = make_mbs(pkg, rng, w // 16, h // 16, 6)
    ctx.me_frame(_params(pkg, R), mbs)
    quants = np.array([pkg.flat_quant(28 + d, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1) for d in (0, 0, 3)], dtype=pkg.QUANT_DTYPE)
    ctx.residual_frame(quants, None)
    recon = ctx.recon_download()
    ctx.recon_to_ref(0)
    with pytest.raises(pkg.JmhipError):
        ctx.recon_download()                                  # the planes now hold the slot's old picture
    with pytest.raises(pkg.JmhipError):
        ctx.recon_to_ref(0)                                   # a second swap would put the old reference back
    ctx.interp_luma(0)
    planes = ctx.download_luma_planes(0)
    assert np.array_equal(planes[0][0][20:20 + h, 20:20 + w], recon[0])   # the slot really holds the reconstruction
    ctx.residual_frame(quants, None)                          # re-arms the recon picture
    ctx.recon_download()
    ctx.close()
