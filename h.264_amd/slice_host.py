"""Host-side helpers of the slice search: a jmhip_slice_params block filled the way the JM binding fills it (integration/jm_shim.c
slice_run), and the small synthetic clip the tools and the smoke test search. Plumbing above the C ABI: nothing here computes a result."""
import numpy as np

from . import jmhip

QP_N = 28


def slice_params(mode, R, nref, lambda_mf, ref_cost1, W, mb_first=0, mb_count=None, H=None, metric=(0, 2, 2), qp_n=QP_N, full_search=2,
                 t8=0, t8_qp=None, cavlc=1, rdopt=0, md_metric=2):
    """SearchMode `mode`, range R, `nref` list-0 references in slots 0..nref-1, every inter mode enabled, JM's default EPZS / UMHexagonS settings
    (jmhip_epzs_setup / jmhip_umhex_setup); t8: Transform8x8Mode with the flat inter 8x8 luma quantiser at t8_qp (jmhip_flat_quant)."""
    lib = jmhip.load_library()
    p = jmhip.SliceParams()
    p.search_mode, p.search_range, p.full_search, p.num_refs = mode, R, full_search, nref
    for r in range(nref):
        p.ref_slot[r] = r
    for m in range(1, 8):
        p.valid[m] = 1
    p.lambda_mf[0], p.lambda_mf[1], p.lambda_mf[2] = lambda_mf
    p.ref_cost1, p.md_metric = ref_cost1, md_metric
    p.metric[0], p.metric[1], p.metric[2] = metric
    p.level_mv_min, p.level_mv_max = -511, 511
    p.rdopt = rdopt
    p.mb_first = mb_first
    p.mb_count = mb_count if mb_count is not None else (W // 16) * (H // 16) - mb_first
    lib.jmhip_epzs_setup(p, R, 2, 3, 2, 1, 1, 1, 0, 1, 2, 2)
    lib.jmhip_umhex_setup(p, 1, 3, qp_n, W)
    if t8:                                           # Transform8x8Mode: the inter 8x8 luma quantiser of the slice (flat matrices, default offsets)
        qp = qp_n if t8_qp is None else t8_qp
        q = jmhip.flat_quant(qp, 342, True)
        p.transform8x8_mode, p.t8_qp, p.t8_cavlc, p.t8_disthres = t8, qp, cavlc, 0
        for k in range(64):
            p.t8_levelscale[k], p.t8_leveloffset[k] = int(q["levelscale"][k]), int(q["leveloffset"][k])
    return p


def synth_clip(rng, W, H, nframes):
    """A smooth moving texture + noise, with a patch that moves differently: frames as uint8 luma planes."""
    yy, xx = np.mgrid[0:H + 64, 0:W + 64]
    base = (np.sin(xx / 7.0) * np.cos(yy / 11.0) * 60 + np.sin((xx + 2 * yy) / 23.0) * 40 + 128 + rng.normal(0, 10, (H + 64, W + 64)))
    out = []
    for f in range(nframes):
        dx, dy = 3 * f, -2 * f
        fr = base[32 + dy:32 + dy + H, 32 + dx:32 + dx + W] + rng.normal(0, 2.5, (H, W))
        fr[H // 3:H // 2, W // 4:W // 2] = base[32 + H // 3 - 3 * f:32 + H // 2 - 3 * f, 32 + W // 4 + f:32 + W // 2 + f]     # a patch moving differently
        out.append(np.clip(np.round(fr), 0, 255).astype(np.uint8))
    return out
