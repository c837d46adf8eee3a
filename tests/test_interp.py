"""getSubImagesLuma / getSubImagesChroma: HIP planes vs the oracle, bit-exact (integer work)."""
import numpy as np
import pytest

from tests import oracle


def _pic(rng, h, w, kind):
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "extremes":   # saturating 6-tap over/undershoot
        return (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    return ((np.sin(xx / 5.0) + np.cos(yy / 7.0)) * 60 + 128 + rng.normal(0, 3, (h, w))).clip(0, 255).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kind", [(64, 48, "noise"), (176, 144, "smooth"), (16, 16, "extremes"),
                                      (272, 32, "extremes"), (1920, 1088, "noise")])
def test_luma_planes_bit_exact(pkg, w, h, kind):
    rng = np.random.default_rng(w * 131 + h)
    Y = _pic(rng, h, w, kind)
    ctx = pkg.Context(w, h, yuv_format=0, max_refs=1)
    ctx.ref_upload(0, Y)
    ctx.interp_luma(0)
    got = ctx.download_luma_planes(0)
    want = oracle.interp_luma(Y)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatch at [py,px,j,i]=%s got %d want %d" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,w,h", [(1, 64, 48), (1, 176, 144), (2, 176, 144), (3, 64, 48), (1, 1920, 1088)])
def test_chroma_planes_bit_exact(pkg, fmt, w, h):
    rng = np.random.default_rng(fmt * 977 + w)
    wc, hc = (w // 2, h // 2) if fmt == 1 else ((w // 2, h) if fmt == 2 else (w, h))
    Y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    U = rng.integers(0, 256, (hc, wc), dtype=np.uint8)
    V = rng.integers(0, 256, (hc, wc), dtype=np.uint8)
    ctx = pkg.Context(w, h, yuv_format=fmt, max_refs=1)
    ctx.ref_upload(0, Y, U, V)
    ctx.interp_chroma(0)
    ctx.interp_chroma(0)      # idempotent: the never-written last row/column must stay zero
    for uv, src in ((0, U), (1, V)):
        got = ctx.download_chroma_planes(0, uv)
        want = oracle.interp_chroma(src, fmt)
        assert got.shape == want.shape
        bad = np.argwhere(got != want)
        assert bad.size == 0, "uv=%d first mismatch at %s" % (uv, bad[0])
        assert not got[:, :, -1, :].any() and not got[:, :, :, -1].any()
    ctx.close()


@pytest.mark.gpu
def test_uint16_imgpel_boundary(pkg):
    """JM hands over unsigned-short samples: upload/download through the 2-byte path."""
    rng = np.random.default_rng(5)
    Y = rng.integers(0, 256, (48, 64), dtype=np.uint16)
    ctx = pkg.Context(64, 48, yuv_format=0)
    ctx.ref_upload(0, Y)
    ctx.interp_luma(0)
    got = ctx.download_luma_planes(0, dtype=np.uint16)
    assert np.array_equal(got, oracle.interp_luma(Y))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,w,h,dtype,gap", [(1, 176, 144, np.uint16, 0), (2, 64, 48, np.uint16, 5), (1, 1920, 1088, np.uint16, 0), (3, 64, 48, np.uint8, 3)])
def test_planes_into_row_pointers(pkg, fmt, w, h, dtype, gap):
    """The JM binding's download (jmhip_ref_download_luma_rows / _chroma_rows): the planes land in the caller's own rows -- JM's imgpel **
    layout, here with `gap` unused samples after every row -- equal to the contiguous download, nothing written past a row's end; called twice
    (the page-locked staging buffer and its events are reused)."""
    rng = np.random.default_rng(fmt * 31 + w)
    wc, hc = (w // 2, h // 2) if fmt == 1 else ((w // 2, h) if fmt == 2 else (w, h))
    ctx = pkg.Context(w, h, yuv_format=fmt, max_refs=2)
    for slot in (0, 1):
        Y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        U = rng.integers(0, 256, (hc, wc), dtype=np.uint8)
        V = rng.integers(0, 256, (hc, wc), dtype=np.uint8)
        ctx.ref_upload(slot, Y, U, V)
        ctx.interp_luma(slot)
        ctx.interp_chroma(slot)
        got = ctx.download_luma_rows(slot, dtype=dtype, gap=gap)
        assert np.array_equal(got[..., :ctx.Wp], ctx.download_luma_planes(slot)) and not got[..., ctx.Wp:].any()
        for uv in (0, 1):
            got = ctx.download_chroma_rows(slot, uv, dtype=dtype, gap=gap)
            assert np.array_equal(got[..., :ctx.Wcp], ctx.download_chroma_planes(slot, uv)) and not got[..., ctx.Wcp:].any()
    ctx.close()
