"""The oracle against golden vectors captured from the REAL JM (tests/golden/*.npz, made by tests/golden/make_golden.py
from oracle/_ref/jm_tap in the build container). Runs everywhere (no GPU, no /root/reference)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tests import oracle

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
IDS = [os.path.basename(g)[:-4] for g in GOLD]


def fnv(a):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    L = oracle.lib()
    L.jmo_fnv1a16.restype = C.c_uint
    L.jmo_fnv1a16.argtypes = [C.c_void_p, C.c_long]
    return int(L.jmo_fnv1a16(a.ctypes.data, a.size))


def recs(z, kind):
    if kind not in z:
        return []
    return [z[kind][i, :z[kind + "_len"][i]] for i in range(len(z[kind]))]


def luma_pic(r):
    w, h = int(r[1]), int(r[2])
    return r[4:4 + w * h].reshape(h, w).astype(np.uint16), w, h


def test_fixtures_present():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_luma_planes_match_jm(path):
    z = np.load(path)
    for r in recs(z, "luma"):
        Y, w, h = luma_pic(r)
        planes = oracle.interp_luma(Y, int(r[3]))
        dig = r[4 + w * h:4 + w * h + 16].astype(np.uint32)
        got = np.array([fnv(planes[p >> 2, p & 3]) for p in range(16)], dtype=np.uint32)
        assert np.array_equal(got, dig)
        rows = r[4 + w * h + 16:].reshape(4, 3, w + 40)        # planes 0,5,10,15; rows 0,17,34
        for k, p in enumerate(range(0, 16, 5)):
            for j in range(3):
                assert np.array_equal(planes[p >> 2, p & 3][(j * 17) % (h + 40)], rows[k, j])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_chroma_planes_match_jm(path):
    z = np.load(path)
    for r in recs(z, "chroma"):
        wc, hc, fmt = int(r[1]), int(r[2]), int(r[3])
        (sx, sy), _ = oracle.chroma_geom(fmt)
        off = 4
        for uv in range(2):
            img = r[off:off + wc * hc].reshape(hc, wc)
            off += wc * hc
            dig = r[off:off + sx * sy].astype(np.uint32)
            off += sx * sy
            planes = oracle.interp_chroma(img, fmt)
            got = np.array([fnv(planes[y, x]) for y in range(sy) for x in range(sx)], dtype=np.uint32)
            assert np.array_equal(got, dig)


def me_setup(z, r):
    """RefPic + oracle params from the 14 recorded globals of a search record."""
    lr = recs(z, "luma")[int(r[0])]
    Y, w, h = luma_pic(lr)
    p = oracle.me_params(rdopt=int(r[1]), is_b_slice=int(r[2]), transform8x8_mode=int(r[4]), metric=(int(r[5]), int(r[6]), int(r[7])))
    p.chroma_me = int(r[3])
    p.apply_weights, p.weight_luma, p.offset_luma = int(r[8]), int(r[9]), int(r[10])
    p.wp_luma_round, p.luma_log_weight_denom = int(r[11]), int(r[12])
    return oracle.RefPic(Y, yuv_format=0), p


_refcache = {}


def cached_setup(path, z, r):
    key = (path, int(r[0]))
    if key not in _refcache:
        _refcache[key] = me_setup(z, r)[0]
    return _refcache[key], me_setup.__wrapped__(z, r) if False else None


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_fullpel_search_matches_jm(path):
    z = np.load(path)
    L = oracle.lib()
    L.jmo_fullpel_search.argtypes = [C.POINTER(oracle.MeParams), C.POINTER(oracle.Ref), C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    cache = {}
    for r in recs(z, "fullpel"):
        if int(r[0]) not in cache:
            cache[int(r[0])] = me_setup(z, r)[0]
        rp, p = cache[int(r[0])], me_setup(z, r)[1] if False else None
        _, p = me_setup.__call__(z, r) if False else (None, None)
        p = oracle.me_params(rdopt=int(r[1]), is_b_slice=int(r[2]), transform8x8_mode=int(r[4]), metric=(int(r[5]), int(r[6]), int(r[7])))
        p.apply_weights, p.weight_luma, p.offset_luma, p.wp_luma_round, p.luma_log_weight_denom = int(r[8]), int(r[9]), int(r[10]), int(r[11]), int(r[12])
        a = r[14:]
        ref0, px, py, bt, pmx, pmy, ix, iy, R, minc, lam, ox, oy, ocost = [int(v) for v in a[:14]]
        blk = np.zeros(768, np.uint16)
        blk[:len(a) - 14] = a[14:]
        mv = np.array([ix, iy], np.int16)
        cost = L.jmo_fullpel_search(C.byref(p), C.byref(rp.ref), blk.ctypes.data, ref0, px, py, bt, pmx, pmy,
                                    mv.ctypes.data, mv[1:].ctypes.data, R, minc, lam)
        assert (int(mv[0]), int(mv[1]), cost) == (ox, oy, ocost)


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_subpel_search_matches_jm(path):
    z = np.load(path)
    L = oracle._setup_search_protos()
    cache = {}
    for r in recs(z, "subpel"):
        if int(r[0]) not in cache:
            cache[int(r[0])] = me_setup(z, r)[0]
        rp = cache[int(r[0])]
        p = oracle.me_params(rdopt=int(r[1]), is_b_slice=int(r[2]), transform8x8_mode=int(r[4]), metric=(int(r[5]), int(r[6]), int(r[7])))
        p.apply_weights, p.weight_luma, p.offset_luma, p.wp_luma_round, p.luma_log_weight_denom = int(r[8]), int(r[9]), int(r[10]), int(r[11]), int(r[12])
        a = r[14:]
        ref0, px, py, bt, pmx, pmy, ix, iy, sp2, sp4, minc, l0, l1, l2, ox, oy, ocost = [int(v) for v in a[:17]]
        blk = np.zeros(768, np.uint16)
        blk[:len(a) - 17] = a[17:]
        mv = np.array([ix, iy], np.int16)
        lam = (C.c_int * 3)(l0, l1, l2)
        cost = L.jmo_subpel_search(C.byref(p), C.byref(rp.ref), blk.ctypes.data, ref0, px, py, bt, pmx, pmy,
                                   mv.ctypes.data, mv[1:].ctypes.data, sp2, sp4, minc, lam)
        assert (int(mv[0]), int(mv[1]), cost) == (ox, oy, ocost)


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_fastfull_search_matches_jm(path):
    z = np.load(path)
    L = oracle._setup_search_protos()
    cache = {}
    for r in recs(z, "fastfull"):
        if int(r[0]) not in cache:
            cache[int(r[0])] = me_setup(z, r)[0]
        rp = cache[int(r[0])]
        p = oracle.me_params(rdopt=int(r[1]), is_b_slice=int(r[2]), transform8x8_mode=int(r[4]), metric=(int(r[5]), int(r[6]), int(r[7])))
        p.apply_weights, p.weight_luma, p.offset_luma, p.wp_luma_round, p.luma_log_weight_denom = int(r[8]), int(r[9]), int(r[10]), int(r[11]), int(r[12])
        a = r[14:]
        ref0, ox0, oy0, px, py, bt, cpx, cpy, pmx, pmy, R, minc, lam, ox, oy, ocost = [int(v) for v in a[:16]]
        mb = np.zeros(768, np.uint16)
        mb[:256] = a[16:16 + 256]
        ff = oracle.FastFull()
        buf = np.zeros(8 * 16 * (2 * R + 1) ** 2, np.int32)
        ff.block_sad = buf.ctypes.data_as(C.POINTER(C.c_int))
        L.jmo_fastfull_setup(C.byref(p), C.byref(rp.ref), mb.ctypes.data, ox0, oy0, cpx, cpy, R, C.byref(ff))
        mv = np.zeros(2, np.int16)
        cost = L.jmo_fastfull_search(C.byref(p), C.byref(ff), ox0, oy0, px, py, bt, pmx, pmy, mv.ctypes.data, mv[1:].ctypes.data, minc, lam)
        assert (int(mv[0]), int(mv[1]), cost) == (ox, oy, ocost)


def quant_from(a, n):
    """(QuantHolder, remaining) from a put_quant block: 9 scalars + 3 tables of n*n."""
    q = np.zeros(1, dtype=[("qp", "<i4"), ("adaptive_rounding", "<i4"), ("adapt_rnd_weight", "<i4"), ("field_scan", "<i4"),
                           ("disthres", "<i4"), ("max_val", "<i4"), ("cavlc", "<i4"), ("img_qp", "<i4"), ("transform8x8_flag", "<i4"),
                           ("levelscale", "<i4", (64,)), ("invlevelscale", "<i4", (64,)), ("leveloffset", "<i4", (64,))])[0]
    for k, name in enumerate(("qp", "adaptive_rounding", "adapt_rnd_weight", "field_scan", "disthres", "max_val", "cavlc", "img_qp", "transform8x8_flag")):
        q[name] = a[k]
    m = n * n
    q["levelscale"][:m], q["invlevelscale"][:m], q["leveloffset"][:m] = a[9:9 + m], a[9 + m:9 + 2 * m], a[9 + 2 * m:9 + 3 * m]
    return q, a[9 + 3 * m:]


def lists_equal(gl, gr, wl, wr):
    nz = np.flatnonzero(np.asarray(wl) == 0)
    k = nz[0] if len(nz) else len(wl) - 1
    return np.array_equal(gl[:k + 1], wl[:k + 1]) and np.array_equal(gr[:k], wr[:k])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_dct_4x4_matches_jm(path):
    z = np.load(path)
    L = oracle.lib()
    vp = C.c_void_p
    L.jmo_dct_4x4.argtypes = [C.POINTER(oracle.Quant), vp, vp, C.c_int, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp]
    for r in recs(z, "dct4"):
        bx, by, intra, cc_in = [int(v) for v in r[:4]]
        q, a = quant_from(r[4:], 4)
        m7 = np.ascontiguousarray(a[:256].reshape(16, 16), np.int32)
        mpr = np.ascontiguousarray(a[256:512].reshape(16, 16), np.uint16)
        o = a[512:]
        ret, cc_out = int(o[0]), int(o[1])
        lev, run, rec, fadj = o[2:19], o[19:36], o[36:52].reshape(4, 4), o[52:52 + 256].reshape(16, 16)
        qh = oracle.QuantHolder(q)
        cost = C.c_int(cc_in)
        gl, gr = np.zeros(17, np.int32), np.zeros(17, np.int32)
        grec, gfa = np.zeros((16, 16), np.uint16), np.zeros((16, 16), np.int32)
        got = L.jmo_dct_4x4(C.byref(qh.c), m7.ctypes.data, mpr.ctypes.data, bx, by, C.byref(cost), gl.ctypes.data, gr.ctypes.data, grec.ctypes.data, gfa.ctypes.data)
        assert (got, cost.value) == (ret, cc_out)
        assert lists_equal(gl, gr, lev, run)
        assert np.array_equal(grec[by:by + 4, bx:bx + 4], rec)
        if q["adaptive_rounding"]:
            assert np.array_equal(gfa[by:by + 4, bx:bx + 4], fadj[by:by + 4, bx:bx + 4])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_dct_8x8_matches_jm(path):
    z = np.load(path)
    L = oracle.lib()
    vp = C.c_void_p
    L.jmo_dct_8x8.argtypes = [C.POINTER(oracle.Quant), vp, vp, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp]
    for r in recs(z, "dct8"):
        b8, intra, cc_in = [int(v) for v in r[:3]]
        q, a = quant_from(r[3:], 8)
        m7 = np.ascontiguousarray(a[:256].reshape(16, 16), np.int32)
        mpr = np.ascontiguousarray(a[256:512].reshape(16, 16), np.uint16)
        o = a[512:]
        ret, cc_out = int(o[0]), int(o[1])
        lr = o[2:2 + 4 * 130].reshape(4, 2, 65)
        rec = o[2 + 520:2 + 520 + 64].reshape(8, 8)
        fadj = o[2 + 520 + 64:2 + 520 + 64 + 256].reshape(16, 16)
        qh = oracle.QuantHolder(q)
        cost = C.c_int(cc_in)
        gl, gr = np.zeros((4, 65), np.int32), np.zeros((4, 65), np.int32)
        grec, gfa = np.zeros((16, 16), np.uint16), np.zeros((16, 16), np.int32)
        got = L.jmo_dct_8x8(C.byref(qh.c), m7.ctypes.data, mpr.ctypes.data, b8, C.byref(cost), gl.ctypes.data, gr.ctypes.data, grec.ctypes.data, gfa.ctypes.data)
        assert (got, cost.value) == (ret, cc_out)
        nlists = 4 if (q["transform8x8_flag"] and q["cavlc"]) else 1
        for k in range(nlists):
            assert lists_equal(gl[k], gr[k], lr[k, 0], lr[k, 1])
        ys, xs = 8 * (b8 >> 1), 8 * (b8 & 1)
        assert np.array_equal(grec[ys:ys + 8, xs:xs + 8], rec)
        if q["adaptive_rounding"]:
            assert np.array_equal(gfa[ys:ys + 8, xs:xs + 8], fadj[ys:ys + 8, xs:xs + 8])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_dct_16x16_matches_jm(path):
    z = np.load(path)
    L = oracle.lib()
    vp = C.c_void_p
    L.jmo_dct_16x16.argtypes = [C.POINTER(oracle.Quant), vp, vp, vp, vp, vp, vp, vp, vp]
    for r in recs(z, "dct16"):
        q, a = quant_from(r[1:], 4)
        cur = np.ascontiguousarray(a[:256].reshape(16, 16), np.uint16)
        pred = np.ascontiguousarray(a[256:512].reshape(16, 16), np.uint16)
        o = a[512:]
        ret = int(o[0])
        dcl, dcr = o[1:18], o[18:35]
        ac = o[35:35 + 16 * 32].reshape(16, 2, 16)
        rec = o[35 + 512:35 + 512 + 256].reshape(16, 16)
        fadj = o[35 + 512 + 256:35 + 512 + 512].reshape(16, 16)
        qh = oracle.QuantHolder(q)
        gdl, gdr = np.zeros(17, np.int32), np.zeros(17, np.int32)
        gal, gar = np.zeros((16, 16), np.int32), np.zeros((16, 16), np.int32)
        grec, gfa = np.zeros((16, 16), np.uint16), np.zeros((16, 16), np.int32)
        got = L.jmo_dct_16x16(C.byref(qh.c), cur.ctypes.data, pred.ctypes.data, gdl.ctypes.data, gdr.ctypes.data, gal.ctypes.data, gar.ctypes.data, grec.ctypes.data, gfa.ctypes.data)
        assert got == ret
        assert lists_equal(gdl, gdr, dcl, dcr)
        for b in range(16):
            assert lists_equal(gal[b], gar[b], ac[b, 0], ac[b, 1])
        assert np.array_equal(grec, rec)
        if q["adaptive_rounding"]:
            # JM only writes AC positions of fadjust in dct_16x16; DC positions keep older values
            mask = np.ones((16, 16), bool)
            mask[::4, ::4] = False
            assert np.array_equal(gfa[mask], fadj[mask])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_dct_chroma_matches_jm(path):
    z = np.load(path)
    L = oracle.lib()
    vp = C.c_void_p
    L.jmo_dct_chroma.argtypes = [C.POINTER(oracle.Quant), C.POINTER(oracle.Quant), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_longlong)]
    for r in recs(z, "dctc"):
        uv, cr_in, fmt = int(r[0]), int(r[1]), int(r[2])
        cbp_in = (int(r[3]) & 0xffffffff) | (int(r[4]) << 32)
        q, a = quant_from(r[5:], 4)
        qdc, a = quant_from(a, 4)
        m7 = np.ascontiguousarray(a[:256].reshape(16, 16), np.int32)
        mpr = np.ascontiguousarray(a[256:512].reshape(16, 16), np.uint16)
        o = a[512:]
        ret = int(o[0])
        cbp_out = (int(o[1]) & 0xffffffff) | (int(o[2]) << 32)
        dcl, dcr = o[3:20], o[20:37]
        ac = o[37:37 + 8 * 32].reshape(8, 2, 16)
        rows, cols = (8, 8) if fmt == 1 else (16, 8)
        rec = o[37 + 256:37 + 256 + rows * cols].reshape(rows, cols)
        fadj = o[37 + 256 + rows * cols:37 + 256 + 2 * rows * cols].reshape(rows, cols)
        qh, qdh = oracle.QuantHolder(q), oracle.QuantHolder(qdc)
        gdl, gdr = np.zeros(17, np.int32), np.zeros(17, np.int32)
        gal, gar = np.zeros((8, 16), np.int32), np.zeros((8, 16), np.int32)
        grec, gfa = np.zeros((16, 16), np.uint16), np.zeros((16, 16), np.int32)
        cbp = C.c_longlong(cbp_in)
        got = L.jmo_dct_chroma(C.byref(qh.c), C.byref(qdh.c), fmt, uv, cr_in, m7.ctypes.data, mpr.ctypes.data, gdl.ctypes.data, gdr.ctypes.data,
                               gal.ctypes.data, gar.ctypes.data, grec.ctypes.data, gfa.ctypes.data, C.byref(cbp))
        assert got == ret
        assert (cbp.value & 0xffffffffffff) == (cbp_out & 0xffffffffffff)
        assert lists_equal(gdl, gdr, dcl, dcr)
        for b in range(4 if fmt == 1 else 8):
            assert lists_equal(gal[b], gar[b], ac[b, 0], ac[b, 1])
        assert np.array_equal(grec[:rows, :cols], rec)
        if q["adaptive_rounding"]:
            mask = np.ones((rows, cols), bool)
            mask[::4, ::4] = False
            assert np.array_equal(gfa[:rows, :cols][mask], fadj[mask])
