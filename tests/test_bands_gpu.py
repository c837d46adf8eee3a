"""The N > 1 data path without RCCL: two "ranks" emulated one after the other on one GPU. Each searches and reconstructs
only its band of macroblock rows (jmhip_me_frame over a row subset, jmhip_residual_frame, jmhip_recon_copy_band into its
send buffer); the bands, concatenated the way the all-gather lays them out, are loaded with jmhip_ref_upload(device pointers)
and must give the same next reference -- integer planes and sub-pel planes -- as one context that coded the whole frame."""
import numpy as np
import pytest
import torch

from tests.test_frame import synth
from tests.test_me import lambda_factors, make_mbs

pytestmark = pytest.mark.gpu


def code(pkg, ctx, cur_dev, ref, rows, mbw, quants, lam, R, band_interp=False):
    from h264_amd.jmhip import ME_MB_DTYPE
    mbs = np.zeros(len(rows) * mbw, dtype=ME_MB_DTYPE)
    k = 0
    for r in rows:
        for x in range(mbw):
            mbs[k]["mb_x"], mbs[k]["mb_y"], mbs[k]["ref_is_0"] = x, r, 1
            mbs[k]["pred_mv"][:] = ((x * 7 + r * 3) % 9 - 4, (x * 5 + r) % 7 - 3)
            k += 1
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt = 0, R, 1
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0], prm.lambda_[1], prm.lambda_[2] = lam
    prm.subpel, prm.partition_mask = 1, (1 << 41) - 1
    ctx.ref_upload(0, *ref)
    if band_interp:      # only the rows this band's search and prediction can reach: own rows +- (range + predictor reach + slack)
        ctx.interp_rows(0, rows[0] * 16 - (R + 16), (rows[-1] + 1) * 16 + (R + 16))
    else:
        ctx.interp_luma(0)
        ctx.interp_chroma(0)
    ctx.cur_bind(*[t.data_ptr() for t in cur_dev])
    ctx.me_frame_async(prm, mbs)
    ctx.residual_frame(quants)
    return mbs


def test_two_bands_rebuild_the_reference(pkg):
    rng = np.random.default_rng(17)
    w, h, R, world = 96, 160, 8, 2
    mbw, mbh = w // 16, h // 16
    cur, ref = synth(rng, w, h, 1)
    dev = torch.device("cuda", 0)
    cur_dev = [torch.from_numpy(p).to(dev) for p in cur]
    lam = lambda_factors(28)
    quants = np.array([pkg.flat_quant(28 + d, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1) for d in (0, 0, 3)], dtype=pkg.QUANT_DTYPE)

    # one context, whole frame
    one = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    code(pkg, one, cur_dev, ref, list(range(mbh)), mbw, quants, lam, R)
    one.recon_to_ref(0)
    one.interp_luma(0)
    one.interp_chroma(0)
    want_luma, want_cb = one.download_luma_planes(0), one.download_chroma_planes(0, 0)
    one.close()

    # two "ranks"
    bandh = -(-mbh // world)
    gbufs = (torch.zeros((world * bandh * 16, w), dtype=torch.uint8, device=dev), torch.zeros((world * bandh * 8, w // 2), dtype=torch.uint8, device=dev),
             torch.zeros((world * bandh * 8, w // 2), dtype=torch.uint8, device=dev))
    for rank in range(world):
        row0, row1, band = pkg.slices.band_rows(mbh, world, rank)
        ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
        code(pkg, ctx, cur_dev, ref, list(range(row0, row1)), mbw, quants, lam, R, band_interp=True)
        sY = torch.zeros((band * 16, w), dtype=torch.uint8, device=dev)
        sU, sV = torch.zeros((band * 8, w // 2), dtype=torch.uint8, device=dev), torch.zeros((band * 8, w // 2), dtype=torch.uint8, device=dev)
        ctx.recon_copy_band(sY.data_ptr(), sU.data_ptr(), sV.data_ptr(), row0, row1 - row0)
        # the exchange is enqueued on the context's own stream, as bench.py does with the RCCL all-gather: no host sync between
        with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_ptr(), device=dev)):
            for g, s, rows_per_mb in zip(gbufs, (sY, sU, sV), (16, 8, 8)):  # what all_gather_into_tensor does with this rank's send buffer
                g[rank * band * rows_per_mb:(rank + 1) * band * rows_per_mb].copy_(s, non_blocking=True)
        ctx.sync()
        ctx.close()
    torch.cuda.synchronize()
    chk = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    chk.ref_upload_device(0, gbufs[0].data_ptr(), gbufs[1].data_ptr(), gbufs[2].data_ptr(), w, w // 2)
    chk.interp_luma(0)
    chk.interp_chroma(0)
    got_luma, got_cb = chk.download_luma_planes(0), chk.download_chroma_planes(0, 0)
    chk.close()
    assert np.array_equal(got_luma, want_luma), "sub-pel planes of the gathered reference differ from the single-context frame"
    assert np.array_equal(got_cb, want_cb)
    assert got_luma[0, 0, 20:20 + h, 20:20 + w].std() > 10      # a real picture, not zeros


def test_interp_rows_matches_the_full_planes_inside_the_range(pkg):
    rng = np.random.default_rng(3)
    w, h = 64, 96
    _, ref = synth(rng, w, h, 2)                       # 4:2:2: chroma rows are luma rows
    full = pkg.Context(w, h, yuv_format=2, max_refs=1, search_range=8)
    full.ref_upload(0, *ref)
    full.interp_luma(0)
    full.interp_chroma(0)
    fl, fc = full.download_luma_planes(0), full.download_chroma_planes(0, 1)
    full.close()
    part = pkg.Context(w, h, yuv_format=2, max_refs=1, search_range=8)
    part.ref_upload(0, *ref)
    part.interp_rows(0, 30, 60)
    pl, pc = part.download_luma_planes(0), part.download_chroma_planes(0, 1)
    part.close()
    assert np.array_equal(pl[:, :, 20 + 30:20 + 60], fl[:, :, 20 + 30:20 + 60])
    assert np.array_equal(pc[:, :, 20 + 30:20 + 60], fc[:, :, 20 + 30:20 + 60])
    assert not pl[:, :, :20].any() and not pl[:, :, 20 + 80:].any()      # far rows untouched (zero-initialised planes)


def test_one_buffer_band_exchange(pkg):
    """jmhip_recon_pack_band / jmhip_ref_unpack_bands: each rank's band as one chunk, `world` chunks scattered into the reference --
    three emulated ranks (the last band partly padding) against the single-context frame."""
    rng = np.random.default_rng(23)
    w, h, R, world = 96, 112, 8, 3                      # 7 macroblock rows: bands of 3, 3, 1 (+2 padding)
    mbw, mbh = w // 16, h // 16
    cur, ref = synth(rng, w, h, 1)
    dev = torch.device("cuda", 0)
    cur_dev = [torch.from_numpy(p).to(dev) for p in cur]
    lam = lambda_factors(28)
    quants = np.array([pkg.flat_quant(28 + d, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1) for d in (0, 0, 3)], dtype=pkg.QUANT_DTYPE)
    one = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    code(pkg, one, cur_dev, ref, list(range(mbh)), mbw, quants, lam, R)
    want = one.recon_download()
    one.close()
    band = -(-mbh // world)
    gbuf = None
    for rank in range(world):
        row0, row1, b = pkg.slices.band_rows(mbh, world, rank)
        ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
        code(pkg, ctx, cur_dev, ref, list(range(row0, row1)), mbw, quants, lam, R, band_interp=True)
        chunk = ctx.band_chunk_bytes(band)
        assert chunk == band * (16 * w + 2 * 8 * (w // 2))
        if gbuf is None:
            gbuf = torch.full((world * chunk,), 0xAB, dtype=torch.uint8, device=dev)
        ctx.recon_pack_band(gbuf[rank * chunk:(rank + 1) * chunk].data_ptr(), rank, band)      # where the all-gather would put it
        ctx.sync()
        ctx.close()
    chk = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=R)
    chk.ref_unpack_bands(0, gbuf.data_ptr(), world, band)
    y, u, v, _, _ = chk.ref_device_planes_ro(0)
    for ptr, wantp in zip((y, u, v), want):
        got = np.zeros_like(wantp)
        chk.copy_from_device(ptr, got)
        assert np.array_equal(got, wantp)
    chk.close()


def test_device_chunk_layout_equals_the_host_mirror(pkg):
    """jmhip_recon_pack_band / jmhip_ref_unpack_bands against h.264_amd/slices.py's host definition of the chunk (what the gloo CPU test exchanges)."""
    rng = np.random.default_rng(23)
    w, h, world = 96, 144, 2                            # 9 macroblock rows in bands of 5: the last band is padded
    mbh = h // 16
    Y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    U = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    V = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    ctx = pkg.Context(w, h, yuv_format=1, max_refs=1, search_range=8)
    ctx.recon_upload(Y, U, V)
    dev = torch.device("cuda", 0)
    band = pkg.slices.band_rows(mbh, world, 0)[2]
    chunk = ctx.band_chunk_bytes(band)
    assert chunk == pkg.slices.chunk_bytes(w, w // 2, 8, band)
    g = torch.zeros(chunk * world, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for r in range(world):
        ctx.recon_pack_band(g[r * chunk:(r + 1) * chunk].data_ptr(), r, band)
    ctx.sync()
    host = g.cpu().numpy()
    for r in range(world):
        want = pkg.slices.pack_band_host(np, Y, U, V, r, band, 8)
        rows_y = min(band * 16, h - r * band * 16)
        assert np.array_equal(host[r * chunk:r * chunk + rows_y * w], want[:rows_y * w]), "luma rows of chunk %d" % r
    ctx.ref_unpack_bands(0, g.data_ptr(), world, band)
    ctx.interp_luma(0)
    planes = ctx.download_luma_planes(0)
    assert np.array_equal(planes[0][0][20:20 + h, 20:20 + w], Y)
    got = pkg.slices.unpack_bands_host(np, host, world, band, h, w, h // 2, w // 2, 8)
    assert all(np.array_equal(a, b) for a, b in zip(got, (Y, U, V)))
    ctx.close()
