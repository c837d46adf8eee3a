"""ctypes wrapper over oracle/liboracle.so (the CPU restatement of JM, test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
PAD = 20
_lib = None


class ChromaGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sub_x", "sub_y", "pad_x", "pad_y", "shift_x", "shift_y", "mask_x", "mask_y",
                                       "mul_x", "mul_y", "mb_cr_size_x", "mb_cr_size_y")]


class Ref(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("Wp", C.c_int), ("Hp", C.c_int),
                ("width_pad", C.c_int), ("height_pad", C.c_int),
                ("luma", C.c_void_p * 16), ("yuv_format", C.c_int), ("cg", ChromaGeom),
                ("Wc", C.c_int), ("Hc", C.c_int), ("Wcp", C.c_int), ("Hcp", C.c_int),
                ("width_pad_cr", C.c_int), ("height_pad_cr", C.c_int), ("cr", (C.c_void_p * 64) * 2)]


class MeParams(C.Structure):
    _fields_ = [("rdopt", C.c_int), ("is_b_slice", C.c_int), ("chroma_me", C.c_int), ("chroma_me_weight", C.c_int),
                ("transform8x8_mode", C.c_int), ("metric", C.c_int * 3), ("apply_weights", C.c_int),
                ("max_val", C.c_int), ("max_val_uv", C.c_int), ("level_mv_min", C.c_int), ("level_mv_max", C.c_int),
                ("weight_luma", C.c_int), ("offset_luma", C.c_int), ("wp_luma_round", C.c_int), ("luma_log_weight_denom", C.c_int),
                ("weight_cr", C.c_int * 2), ("offset_cr", C.c_int * 2), ("wp_chroma_round", C.c_int), ("chroma_log_weight_denom", C.c_int)]


class Dist(C.Structure):
    _fields_ = [("ref", C.POINTER(Ref)), ("umv", C.c_int), ("chroma_me", C.c_int), ("chroma_me_weight", C.c_int),
                ("test8x8", C.c_int), ("max_val", C.c_int), ("max_val_uv", C.c_int),
                ("weight_luma", C.c_int), ("offset_luma", C.c_int), ("wp_luma_round", C.c_int), ("luma_log_weight_denom", C.c_int),
                ("weight_cr", C.c_int * 2), ("offset_cr", C.c_int * 2), ("wp_chroma_round", C.c_int), ("chroma_log_weight_denom", C.c_int)]


def build():
    r = subprocess.run(["make", "-C", ODIR, "-j4", "liboracle.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building liboracle.so failed:\n" + r.stderr[-3000:])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ODIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, ip = C.c_void_p, C.c_int
        L.jmo_interp_luma.argtypes = [vp, ip, ip, ip, ip, vp]
        L.jmo_interp_luma.restype = None
        L.jmo_interp_chroma.argtypes = [vp, ip, ip, ip, ip, vp]
        L.jmo_interp_chroma.restype = None
        L.jmo_ref_init.argtypes = [C.POINTER(Ref), ip, ip, ip, vp, vp, vp]
        L.jmo_ref_init.restype = None
        L.jmo_block_search_full.argtypes = [C.POINTER(MeParams), C.POINTER(Ref), vp, ip, ip, ip, ip, ip, ip, ip,
                                            C.POINTER(ip), vp, vp, C.POINTER(ip)]
        L.jmo_fastfull_setup.argtypes = None
        L.jmo_sad.argtypes = [C.POINTER(Dist), vp, ip, ip, ip, ip, ip]
        L.jmo_satd.argtypes = [C.POINTER(Dist), vp, ip, ip, ip, ip, ip]
        _lib = L
    return _lib


def chroma_geom(yuv_format):
    return {1: ((8, 8), (10, 10)), 2: ((8, 4), (10, 20)), 3: ((4, 4), (20, 20))}[yuv_format]


def interp_luma(img, max_val=255):
    """img: (H, W) -> (4, 4, H+40, W+40) uint16, JM's imgY_sub."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    H, W = img.shape
    out = np.zeros((4, 4, H + 2 * PAD, W + 2 * PAD), dtype=np.uint16)
    lib().jmo_interp_luma(img.ctypes.data, W, H, W, max_val, out.ctypes.data)
    return out


def interp_chroma(img, yuv_format):
    """img: (Hc, Wc) -> (sub_y, sub_x, Hc+2pad_y, Wc+2pad_x) uint16, JM's imgUV_sub[uv] (calloc zeros kept)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    Hc, Wc = img.shape
    (sx, sy), (px, py) = chroma_geom(yuv_format)
    out = np.zeros((sy, sx, Hc + 2 * py, Wc + 2 * px), dtype=np.uint16)
    lib().jmo_interp_chroma(img.ctypes.data, Wc, Hc, Wc, yuv_format, out.ctypes.data)
    return out


class RefPic:
    """A reference picture with all sub-pel planes built by the oracle."""

    def __init__(self, Y, U=None, V=None, yuv_format=1):
        self.Y = np.ascontiguousarray(Y, dtype=np.uint16)
        self.H, self.W = self.Y.shape
        self.yuv_format = yuv_format
        self.luma = interp_luma(self.Y)
        self.cb = interp_chroma(U, yuv_format) if U is not None else None
        self.cr = interp_chroma(V, yuv_format) if V is not None else None
        self.ref = Ref()
        lib().jmo_ref_init(C.byref(self.ref), self.W, self.H, yuv_format, self.luma.ctypes.data,
                           self.cb.ctypes.data if self.cb is not None else None,
                           self.cr.ctypes.data if self.cr is not None else None)


def me_params(rdopt=1, is_b_slice=0, transform8x8_mode=0, metric=(0, 2, 2), level_mv=(-511, 511), chroma_me=0, chroma_me_weight=1):
    p = MeParams()
    p.rdopt, p.is_b_slice, p.transform8x8_mode = rdopt, is_b_slice, transform8x8_mode
    p.metric[0], p.metric[1], p.metric[2] = metric
    p.chroma_me, p.chroma_me_weight = chroma_me, chroma_me_weight
    p.max_val = p.max_val_uv = 255
    p.level_mv_min, p.level_mv_max = level_mv
    return p


BLOCK_SIZE = {1: (16, 16), 2: (16, 8), 3: (8, 16), 4: (8, 8), 5: (8, 4), 6: (4, 8), 7: (4, 4)}


def chroma_shifts(yuv_format):
    """luma -> chroma block size shifts (x, y)"""
    return (1 if yuv_format in (1, 2) else 0), (1 if yuv_format == 1 else 0)


def orig_block(cur, pic_x, pic_y, bsx, bsy, cur_uv=None, yuv_format=1):
    """orig_pic of BlockMotionSearch (mv-search.c:605-630): the luma block, then the Cb block at 256 and the Cr block at 512."""
    orig = np.zeros(768, dtype=np.uint16)
    orig[: bsx * bsy] = np.asarray(cur[pic_y:pic_y + bsy, pic_x:pic_x + bsx], dtype=np.uint16).reshape(-1)
    if cur_uv is not None:
        sx, sy = chroma_shifts(yuv_format)
        for k in range(2):
            blk = np.asarray(cur_uv[k][pic_y >> sy:(pic_y + bsy) >> sy, pic_x >> sx:(pic_x + bsx) >> sx], dtype=np.uint16).reshape(-1)
            orig[256 << k:(256 << k) + blk.size] = blk
    return orig


def block_search_full(p, refpic, cur, pic_x, pic_y, blocktype, pred, R, lam, ref_is_0=1, cur_uv=None):
    """BlockMotionSearch chain for SearchMode=-1 -> (mv_qpel, cost, mv_int, cost_int)."""
    bsx, bsy = BLOCK_SIZE[blocktype]
    orig = orig_block(cur, pic_x, pic_y, bsx, bsy, cur_uv, refpic.yuv_format)
    lam_a = (C.c_int * 3)(*lam)
    mv = np.zeros(2, dtype=np.int16)
    mvi = np.zeros(2, dtype=np.int16)
    ci = C.c_int()
    cost = lib().jmo_block_search_full(C.byref(p), C.byref(refpic.ref), orig.ctypes.data, ref_is_0, pic_x, pic_y, blocktype,
                                       int(pred[0]), int(pred[1]), R, lam_a, mv.ctypes.data, mvi.ctypes.data, C.byref(ci))
    return (int(mv[0]), int(mv[1])), cost, (int(mvi[0]), int(mvi[1])), ci.value


# ------------------------------------------------------------------ frame-level ME reference (loops in Python)

class FastFull(C.Structure):
    _fields_ = [("search_range", C.c_int), ("max_pos", C.c_int), ("center_x", C.c_int), ("center_y", C.c_int),
                ("pos_00", C.c_int), ("block_sad", C.POINTER(C.c_int))]


PARTS = ([(1, 0, 0, 4, 4)] + [(2, 0, 2 * k, 4, 2) for k in range(2)] + [(3, 2 * k, 0, 2, 4) for k in range(2)]
         + [(4, 2 * (b & 1), 2 * (b >> 1), 2, 2) for b in range(4)]
         + [(5, 2 * (b & 1), 2 * (b >> 1) + k, 2, 1) for b in range(4) for k in range(2)]
         + [(6, 2 * (b & 1) + k, 2 * (b >> 1), 1, 2) for b in range(4) for k in range(2)]
         + [(7, 2 * (b & 1) + (k & 1), 2 * (b >> 1) + (k >> 1), 1, 1) for b in range(4) for k in range(4)])


def _setup_search_protos():
    L = lib()
    ip, vp = C.c_int, C.c_void_p
    L.jmo_fastfull_setup.argtypes = [C.POINTER(MeParams), C.POINTER(Ref), vp, ip, ip, ip, ip, ip, C.POINTER(FastFull)]
    L.jmo_fastfull_setup.restype = None
    L.jmo_fastfull_search.argtypes = [C.POINTER(MeParams), C.POINTER(FastFull), ip, ip, ip, ip, ip, ip, ip, vp, vp, ip, ip]
    L.jmo_subpel_search.argtypes = [C.POINTER(MeParams), C.POINTER(Ref), vp, ip, ip, ip, ip, ip, ip, vp, vp, ip, ip, ip, C.POINTER(ip)]
    return L


def me_frame(p, refpics, cur, mbs, mode, R, lam, subpel=True, mask=(1 << 41) - 1, cur_uv=None):
    """Reference for jmhip_me_frame: returns dict of arrays mv, cost, mv_int, cost_int shaped like the ABI result.
    cur_uv: (U, V) of the current picture when p.chroma_me is set."""
    L = _setup_search_protos()
    start_hp = 0 if (p.chroma_me == 1 or p.metric[0] != p.metric[1]) else 1          # mv-search.c:396
    if cur_uv is not None:
        cur_uv = [np.ascontiguousarray(c, dtype=np.uint16) for c in cur_uv]
    n = len(mbs)
    out = {"mv": np.zeros((n, 41, 2), np.int16), "cost": np.full((n, 41), -1, np.int32),
           "mv_int": np.zeros((n, 41, 2), np.int16), "cost_int": np.full((n, 41), -1, np.int32)}
    cur16 = np.ascontiguousarray(cur, dtype=np.uint16)
    lam_a = (C.c_int * 3)(*lam)
    for i, mb in enumerate(mbs):
        rp = refpics[int(mb["ref"])]
        ox, oy = int(mb["mb_x"]) * 16, int(mb["mb_y"]) * 16
        ff = None
        if mode == 0:
            mbpix = np.zeros(768, np.uint16)
            mbpix[:256] = cur16[oy:oy + 16, ox:ox + 16].reshape(-1)
            if cur_uv is not None:                      # me_fullfast.c:577-587: Cb rows, then Cr rows, right behind the luma block
                sx, sy = chroma_shifts(rp.yuv_format)
                o = 256
                for k in range(2):
                    blk = cur_uv[k][oy >> sy:(oy + 16) >> sy, ox >> sx:(ox + 16) >> sx].reshape(-1)
                    mbpix[o:o + blk.size] = blk
                    o += blk.size
            ff = FastFull()
            buf = np.zeros(8 * 16 * (2 * R + 1) ** 2, np.int32)
            ff.block_sad = buf.ctypes.data_as(C.POINTER(C.c_int))
            L.jmo_fastfull_setup(C.byref(p), C.byref(rp.ref), mbpix.ctypes.data, ox, oy,
                                 int(mb["pred_mv"][0][0]), int(mb["pred_mv"][0][1]), R, C.byref(ff))
        for q, (bt, x4, y4, w4, h4) in enumerate(PARTS):
            if not (mask >> q) & 1:
                continue
            px, py = ox + 4 * x4, oy + 4 * y4
            pred = (int(mb["pred_mv"][q][0]), int(mb["pred_mv"][q][1]))
            if mode == -1 and subpel:
                mv, cost, mvi, ci = block_search_full(p, rp, cur16, px, py, bt, pred, R, lam, int(mb["ref_is_0"]), cur_uv)
            else:
                bsx, bsy = 4 * w4, 4 * h4
                orig = orig_block(cur16, px, py, bsx, bsy, cur_uv, rp.yuv_format)
                mvs = np.zeros(2, np.int16)
                if mode == 0:
                    ci = L.jmo_fastfull_search(C.byref(p), C.byref(ff), ox, oy, px, py, bt, pred[0], pred[1],
                                               mvs.ctypes.data, mvs[1:].ctypes.data, 2147483647, lam[0])
                else:
                    L.jmo_search_center.argtypes = [C.POINTER(MeParams), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
                    L.jmo_search_center(C.byref(p), pred[0], pred[1], R, mvs.ctypes.data, mvs[1:].ctypes.data)
                    L.jmo_fullpel_search.argtypes = [C.POINTER(MeParams), C.POINTER(Ref), C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
                    ci = L.jmo_fullpel_search(C.byref(p), C.byref(rp.ref), orig.ctypes.data, int(mb["ref_is_0"]), px, py, bt,
                                              pred[0], pred[1], mvs.ctypes.data, mvs[1:].ctypes.data, R, 2147483647, lam[0])
                mvi = (int(mvs[0]), int(mvs[1]))
                if subpel:
                    mvq = np.array([mvi[0] << 2, mvi[1] << 2], np.int16)
                    cost = L.jmo_subpel_search(C.byref(p), C.byref(rp.ref), orig.ctypes.data, int(mb["ref_is_0"]), px, py, bt,
                                               pred[0], pred[1], mvq.ctypes.data, mvq[1:].ctypes.data, 9, 9,
                                               ci if start_hp else 2147483647, lam_a)       # mv-search.c:785-788
                    mv = (int(mvq[0]), int(mvq[1]))
                else:
                    mv, cost = (mvi[0] << 2, mvi[1] << 2), ci
            out["mv"][i, q] = mv
            out["cost"][i, q] = cost
            out["mv_int"][i, q] = mvi
            out["cost_int"][i, q] = ci
    return out


# ------------------------------------------------------------------ transform / quant reference

class Quant(C.Structure):
    _fields_ = [("qp", C.c_int), ("levelscale", C.POINTER(C.c_int)), ("invlevelscale", C.POINTER(C.c_int)),
                ("leveloffset", C.POINTER(C.c_int)), ("adaptive_rounding", C.c_int), ("adapt_rnd_weight", C.c_int),
                ("field_scan", C.c_int), ("disthres", C.c_int), ("max_val", C.c_int), ("cavlc", C.c_int),
                ("img_qp", C.c_int), ("transform8x8_flag", C.c_int)]


class QuantHolder:
    """Keeps the table arrays alive next to the C struct. q: a jmhip QUANT_DTYPE record (same fields)."""

    def __init__(self, q):
        self.ls = np.ascontiguousarray(q["levelscale"], dtype=np.int32)
        self.ils = np.ascontiguousarray(q["invlevelscale"], dtype=np.int32)
        self.lo = np.ascontiguousarray(q["leveloffset"], dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self.c = Quant(int(q["qp"]), self.ls.ctypes.data_as(ip), self.ils.ctypes.data_as(ip), self.lo.ctypes.data_as(ip),
                       int(q["adaptive_rounding"]), int(q["adapt_rnd_weight"]), int(q["field_scan"]), int(q["disthres"]),
                       int(q["max_val"]), int(q["cavlc"]), int(q["img_qp"]), int(q["transform8x8_flag"]))


def flat_tables(qp, offset11, is8x8=False):
    n = 64 if is8x8 else 16
    ls, ils, lo = (np.zeros(64, np.int32) for _ in range(3))
    f = lib().jmo_flat_tables8x8 if is8x8 else lib().jmo_flat_tables4x4
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = None
    f(qp, offset11, ls.ctypes.data, ils.ctypes.data, lo.ctypes.data)
    return ls, ils, lo, n


def tq_reference(kind, quants, jobs, yuv_format=1):
    """Runs the oracle's dct_* on every job; returns a dict of arrays shaped like jmhip TQ_RESULT_DTYPE fields."""
    L = lib()
    n = len(jobs)
    out = {"levels": np.zeros((n, 16, 17), np.int32), "runs": np.zeros((n, 16, 17), np.int32),
           "levels8": np.zeros((n, 4, 65), np.int32), "runs8": np.zeros((n, 4, 65), np.int32),
           "dc_levels": np.zeros((n, 17), np.int32), "dc_runs": np.zeros((n, 17), np.int32),
           "recon": np.zeros((n, 16, 16), np.uint8), "fadjust": np.zeros((n, 16, 16), np.int32),
           "coeff_cost": np.zeros((n, 16), np.int32), "nonzero": np.zeros((n, 16), np.int32),
           "ret": np.zeros(n, np.int32), "cbp_blk": np.zeros(n, np.int64), "cbp_clear": np.zeros(n, np.int64)}
    holders = [QuantHolder(q) for q in quants]
    vp = C.c_void_p
    for t, job in enumerate(jobs):
        q = holders[int(job["quant"])]
        src = job["src"].astype(np.int32)
        pred16 = np.ascontiguousarray(job["pred"], dtype=np.uint16)
        m7 = np.ascontiguousarray(src - job["pred"].astype(np.int32), dtype=np.int32)
        recon = np.zeros((16, 16), np.uint16)
        fadj = np.zeros((16, 16), np.int32)
        if kind == "luma4x4":
            for blk in range(16):
                b8, b4 = blk >> 2, blk & 3
                bx, by = 8 * (b8 & 1) + 4 * (b4 & 1), 8 * (b8 >> 1) + 4 * (b4 >> 1)
                cost = C.c_int(0)
                lev, run = np.zeros(17, np.int32), np.zeros(17, np.int32)
                L.jmo_dct_4x4.argtypes = [C.POINTER(Quant), vp, vp, C.c_int, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp]
                nz = L.jmo_dct_4x4(C.byref(q.c), m7.ctypes.data, pred16.ctypes.data, bx, by, C.byref(cost),
                                   lev.ctypes.data, run.ctypes.data, recon.ctypes.data, fadj.ctypes.data)
                out["levels"][t, blk], out["runs"][t, blk] = lev, run
                out["coeff_cost"][t, blk], out["nonzero"][t, blk] = cost.value, nz
        elif kind == "luma8x8":
            for b8 in range(4):
                cost = C.c_int(0)
                lev, run = np.zeros((4, 65), np.int32), np.zeros((4, 65), np.int32)
                L.jmo_dct_8x8.argtypes = [C.POINTER(Quant), vp, vp, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp]
                nz = L.jmo_dct_8x8(C.byref(q.c), m7.ctypes.data, pred16.ctypes.data, b8, C.byref(cost),
                                   lev.ctypes.data, run.ctypes.data, recon.ctypes.data, fadj.ctypes.data)
                if q.c.transform8x8_flag and q.c.cavlc:
                    out["levels"][t, 4 * b8:4 * b8 + 4], out["runs"][t, 4 * b8:4 * b8 + 4] = lev[:, :17], run[:, :17]
                else:
                    out["levels8"][t, b8], out["runs8"][t, b8] = lev[0], run[0]
                out["coeff_cost"][t, b8], out["nonzero"][t, b8] = cost.value, nz
        elif kind == "luma16x16":
            cur16 = np.ascontiguousarray(job["src"], dtype=np.uint16)
            acl, acr = np.zeros((16, 16), np.int32), np.zeros((16, 16), np.int32)
            dcl, dcr = np.zeros(17, np.int32), np.zeros(17, np.int32)
            L.jmo_dct_16x16.argtypes = [C.POINTER(Quant), vp, vp, vp, vp, vp, vp, vp, vp]
            out["ret"][t] = L.jmo_dct_16x16(C.byref(q.c), cur16.ctypes.data, pred16.ctypes.data, dcl.ctypes.data, dcr.ctypes.data,
                                            acl.ctypes.data, acr.ctypes.data, recon.ctypes.data, fadj.ctypes.data)
            out["levels"][t, :, :16], out["runs"][t, :, :16] = acl, acr
            out["dc_levels"][t], out["dc_runs"][t] = dcl, dcr
        else:
            qdc = holders[int(job["quant_dc"])]
            acl, acr = np.zeros((8, 16), np.int32), np.zeros((8, 16), np.int32)
            dcl, dcr = np.zeros(17, np.int32), np.zeros(17, np.int32)
            cbp = C.c_longlong(0)
            L.jmo_dct_chroma.argtypes = [C.POINTER(Quant), C.POINTER(Quant), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                         C.POINTER(C.c_longlong)]
            # run twice to separate set bits from cleared bits: start from all-ones and from zero
            m7b = m7.copy()
            out["ret"][t] = L.jmo_dct_chroma(C.byref(q.c), C.byref(qdc.c), yuv_format, int(job["uv"]), int(job["cr_cbp_in"]),
                                             m7.ctypes.data, pred16.ctypes.data, dcl.ctypes.data, dcr.ctypes.data,
                                             acl.ctypes.data, acr.ctypes.data, recon.ctypes.data, fadj.ctypes.data, C.byref(cbp))
            ones = C.c_longlong(-1)
            d2, d3, a2, a3 = np.zeros(17, np.int32), np.zeros(17, np.int32), np.zeros((8, 16), np.int32), np.zeros((8, 16), np.int32)
            r2, f2 = np.zeros((16, 16), np.uint16), np.zeros((16, 16), np.int32)
            L.jmo_dct_chroma(C.byref(q.c), C.byref(qdc.c), yuv_format, int(job["uv"]), int(job["cr_cbp_in"]),
                             m7b.ctypes.data, pred16.ctypes.data, d2.ctypes.data, d3.ctypes.data, a2.ctypes.data, a3.ctypes.data,
                             r2.ctypes.data, f2.ctypes.data, C.byref(ones))
            out["cbp_blk"][t] = cbp.value
            out["cbp_clear"][t] = ~ones.value
            out["levels"][t, :8, :16], out["runs"][t, :8, :16] = acl, acr
            out["dc_levels"][t], out["dc_runs"][t] = dcl, dcr
        out["recon"][t] = recon.astype(np.uint8)
        out["fadjust"][t] = fadj
    return out


# ------------------------------------------------------------------ frame stage reference (MC -> residual -> TQ -> thresholds)

def pick_modes(cost):
    """Smallest summed motion cost, ties to the lower mode number (the device's stand-in for mode decision)."""
    n = cost.shape[0]
    modes = np.zeros(n, dtype=[("mode", "i1"), ("b8mode", "i1", (4,)), ("pad", "i1", (3,))])
    for i in range(n):
        c = cost[i].astype(np.int64)
        c8 = 0
        for b in range(4):
            cand = [(c[5 + b], 4), (c[9 + 2 * b] + c[10 + 2 * b], 5), (c[17 + 2 * b] + c[18 + 2 * b], 6),
                    (c[25 + 4 * b:29 + 4 * b].sum(), 7)]
            best = min(cand, key=lambda t: (t[0], t[1]))
            modes[i]["b8mode"][b] = best[1]
            c8 += best[0]
        cand = [(c[0], 1), (c[1] + c[2], 2), (c[3] + c[4], 3), (c8, 8)]
        modes[i]["mode"] = min(cand, key=lambda t: (t[0], t[1]))[1]
    return modes


def covering_partition(m, x4, y4):
    b8 = 2 * (y4 >> 1) + (x4 >> 1)
    if m["mode"] == 1:
        return 0
    if m["mode"] == 2:
        return 1 + (y4 >> 1)
    if m["mode"] == 3:
        return 3 + (x4 >> 1)
    bm = m["b8mode"][b8]
    if bm == 4:
        return 5 + b8
    if bm == 5:
        return 9 + 2 * b8 + (y4 & 1)
    if bm == 6:
        return 17 + 2 * b8 + (x4 & 1)
    return 25 + 4 * b8 + 2 * (y4 & 1) + (x4 & 1)


def residual_frame(refpic, cur, mbs, mv, modes, quants3, job_dtype, yuv_format=1, blk_ref=None, wp=None, bi=None, bw=None):
    """Reference for jmhip_residual_frame. refpic: RefPic with chroma planes; cur = (Y,U,V); mv: (n,41,2) quarter-pel.
    blk_ref (n,4): refpic is then a LIST of RefPic and each 8x8 block predicts from refpic[blk_ref[i][b8]] (LumaPrediction's l0_ref_idx,
    macroblock.c:836). wp: explicit weighted uni-prediction (macroblock.c:899-903 luma, :1895-1898 chroma), the dict Context.frame_wp_set takes.
    bi (MB_BIPRED_DTYPE, n): B macroblocks -- per 8x8 block p_dir 0 / 1 / 2, the list-1 reference slot, list-1 vectors per 4x4 block; bw: the
    second list's weights (Context.frame_bipred_set). Mixes: macroblock.c:880-940 (luma), :1768-1830 (chroma, incl. the luma denominator in
    the bi-predictive chroma shift :1781)."""
    Y, U, V = cur
    n = len(mbs)
    refs = refpic if isinstance(refpic, (list, tuple)) else [refpic]
    refpic = refs[0]

    def mix(pdir, s0, s1, comp, v0, v1):
        """LumaPrediction / ChromaPrediction4x4's combination of the list-0 fetch v0 and the list-1 fetch v1 (arrays)."""
        if wp is None:
            return ((v0.astype(np.int64) + v1 + 1) >> 1) if pdir == 2 else (v1 if pdir else v0)
        rnd, den = (wp["luma_round"], wp["luma_denom"]) if comp == 0 else (wp["chroma_round"], wp["chroma_denom"])
        if pdir == 2:
            off = (int(wp["offset"][s0][comp]) + int(bw["offset1"][s1][comp]) + 1) >> 1
            return np.clip(((int(bw["w0"][s0][s1][comp]) * v0.astype(np.int64) + int(bw["w1"][s0][s1][comp]) * v1.astype(np.int64) + 2 * rnd) >> (wp["luma_denom"] + 1)) + off, 0, 255)
        if pdir == 0:
            return np.clip(((int(wp["weight"][s0][comp]) * v0.astype(np.int64) + rnd) >> den) + int(wp["offset"][s0][comp]), 0, 255)
        return np.clip(((int(bw["weight1"][s1][comp]) * v1.astype(np.int64) + rnd) >> den) + int(bw["offset1"][s1][comp]), 0, 255)

    def weigh(v, slot, comp):
        if wp is None:
            return v
        rnd, den = (wp["luma_round"], wp["luma_denom"]) if comp == 0 else (wp["chroma_round"], wp["chroma_denom"])
        return np.clip(((int(wp["weight"][slot][comp]) * v.astype(np.int64) + rnd) >> den) + int(wp["offset"][slot][comp]), 0, 255)

    (sx, sy), (px, py) = chroma_geom(yuv_format)
    shift_x, shift_y = (3, 3) if yuv_format == 1 else (3, 2)
    mcw, mch = (8, 8) if yuv_format == 1 else (8, 16)
    Wp, Hp = refpic.W + 40, refpic.H + 40
    Wcp, Hcp = refpic.cb.shape[3], refpic.cb.shape[2]
    jobs_y = np.zeros(n, dtype=job_dtype)
    jobs_c = np.zeros(2 * n, dtype=job_dtype)
    for i, mb in enumerate(mbs):
        mbx, mby = int(mb["mb_x"]), int(mb["mb_y"])
        mv4 = np.zeros((4, 4, 2), int)
        for y4 in range(4):
            for x4 in range(4):
                p = covering_partition(modes[i], x4, y4)
                mv4[y4, x4] = mv[i, p]
                # with the 8x8 transform LumaPrediction runs per 8x8 block (macroblock.c:1143): its origin is what UMVLine4X clamps
                t8 = int(modes[i]["pad"][0])
                ox4, oy4 = ((x4 & 1) * 4, (y4 & 1) * 4) if t8 else (0, 0)
                xq = ((mbx * 16 + 4 * x4 - ox4) << 2) + 80 + int(mv[i, p, 0])
                yq = ((mby * 16 + 4 * y4 - oy4) << 2) + 80 + int(mv[i, p, 1])
                xpos = min(max(xq >> 2, 0), Wp - 17) + ox4
                ypos = min(max(yq >> 2, 0), Hp - 17) + oy4
                b8 = 2 * (y4 >> 1) + (x4 >> 1)
                slot = int(blk_ref[i][b8]) if blk_ref is not None else 0
                v0 = refs[slot].luma[yq & 3, xq & 3, ypos:ypos + 4, xpos:xpos + 4]
                if bi is None:
                    jobs_y[i]["pred"][4 * y4:4 * y4 + 4, 4 * x4:4 * x4 + 4] = weigh(v0, slot, 0)
                else:
                    pdir, s1 = int(bi[i]["pdir"][b8]), int(bi[i]["ref1"][b8])
                    xq1 = ((mbx * 16 + 4 * x4 - ox4) << 2) + 80 + int(bi[i]["mv1"][4 * y4 + x4][0])
                    yq1 = ((mby * 16 + 4 * y4 - oy4) << 2) + 80 + int(bi[i]["mv1"][4 * y4 + x4][1])
                    xp1 = min(max(xq1 >> 2, 0), Wp - 17) + ox4
                    yp1 = min(max(yq1 >> 2, 0), Hp - 17) + oy4
                    v1 = refs[s1].luma[yq1 & 3, xq1 & 3, yp1:yp1 + 4, xp1:xp1 + 4] if pdir else v0
                    jobs_y[i]["pred"][4 * y4:4 * y4 + 4, 4 * x4:4 * x4 + 4] = mix(pdir, slot, s1, 0, v0, v1)
        jobs_y[i]["src"] = Y[mby * 16:mby * 16 + 16, mbx * 16:mbx * 16 + 16]
        for uv, C_ in enumerate((U, V)):
            jc = jobs_c[2 * i + uv]
            jc["quant"], jc["quant_dc"], jc["uv"] = 1, 2, uv
            for j in range(mch):
                for ic in range(0, mcw, 2):
                    by4, bx4 = j >> (4 - shift_y), ic >> (4 - shift_x)
                    ii = ((ic + mbx * mcw) << shift_x) + 80 + int(mv4[by4, bx4, 0])
                    jj = ((j + mby * mch) << shift_y) + 80 + int(mv4[by4, bx4, 1])
                    xpos = min(max(ii >> shift_x, 0), Wcp - 1 - mcw)
                    ypos = min(max(jj >> shift_y, 0), Hcp - 1 - mch)
                    b8 = 2 * (by4 >> 1) + (bx4 >> 1)
                    slot = int(blk_ref[i][b8]) if blk_ref is not None else 0
                    planes = refs[slot].cr if uv else refs[slot].cb
                    v0 = planes[jj & ((1 << shift_y) - 1), ii & ((1 << shift_x) - 1), ypos, xpos:xpos + 2]
                    if bi is None:
                        jc["pred"][j, ic:ic + 2] = weigh(v0, slot, uv + 1)
                    else:
                        pdir, s1 = int(bi[i]["pdir"][b8]), int(bi[i]["ref1"][b8])
                        m1 = bi[i]["mv1"][4 * by4 + bx4]
                        ii1 = ((ic + mbx * mcw) << shift_x) + 80 + int(m1[0])
                        jj1 = ((j + mby * mch) << shift_y) + 80 + int(m1[1])
                        xp1 = min(max(ii1 >> shift_x, 0), Wcp - 1 - mcw)
                        yp1 = min(max(jj1 >> shift_y, 0), Hcp - 1 - mch)
                        pl1 = refs[s1].cr if uv else refs[s1].cb
                        v1 = pl1[jj1 & ((1 << shift_y) - 1), ii1 & ((1 << shift_x) - 1), yp1, xp1:xp1 + 2] if pdir else v0
                        jc["pred"][j, ic:ic + 2] = mix(pdir, slot, s1, uv + 1, v0, v1)
            jc["src"][:mch, :mcw] = C_[mby * mch:mby * mch + mch, mbx * mcw:mbx * mcw + mcw]
    ry = tq_reference("luma4x4", quants3, jobs_y)
    t8 = np.array([int(m["pad"][0]) for m in modes], bool)
    if t8.any():
        jobs8 = jobs_y.copy()
        jobs8["quant"] = 3
        r8 = tq_reference("luma8x8", quants3, jobs8)
        for k in ry:
            ry[k][t8] = r8[k][t8]
    rc = tq_reference("chroma", quants3, jobs_c, yuv_format=yuv_format)
    cbp = np.zeros(n, np.int32)
    cbp_blk = np.zeros(n, np.int64)
    recY = np.zeros_like(Y)
    recU, recV = np.zeros_like(U), np.zeros_like(V)
    for i, mb in enumerate(mbs):
        mbx, mby = int(mb["mb_x"]), int(mb["mb_y"])
        c, cb, total = 0, 0, 0
        rec = ry["recon"][i].copy()
        pred = jobs_y[i]["pred"]
        for b8 in range(4):
            if t8[i]:                                       # macroblock.c:1181-1188
                cost = int(ry["coeff_cost"][i, b8])
                if ry["nonzero"][i, b8]:
                    c |= 1 << b8
                    cb |= 51 << (4 * b8 - 2 * (b8 & 1))
            else:
                cost = int(ry["coeff_cost"][i, 4 * b8:4 * b8 + 4].sum())
            for b4 in range(4):
                if not t8[i] and ry["nonzero"][i, 4 * b8 + b4]:
                    c |= 1 << b8
                    cb |= 1 << ((2 * (b8 & 1) + (b4 & 1)) + 4 * (2 * (b8 >> 1) + (b4 >> 1)))
            if cost <= 4:
                cost = 0
                c &= 63 - (1 << b8)
                cb &= ~(51 << (4 * b8 - 2 * (b8 & 1)))
                ys, xs = 8 * (b8 >> 1), 8 * (b8 & 1)
                rec[ys:ys + 8, xs:xs + 8] = pred[ys:ys + 8, xs:xs + 8]
            total += cost
        if total <= 5:
            c &= 0xfffff0
            cb &= 0xff0000
            rec = pred.copy()
        for uv in range(2):
            cb = (cb & ~int(rc["cbp_clear"][2 * i + uv])) | int(rc["cbp_blk"][2 * i + uv] & ~rc["cbp_clear"][2 * i + uv])
        c += max(int(rc["ret"][2 * i]), int(rc["ret"][2 * i + 1])) << 4
        cbp[i], cbp_blk[i] = c, cb
        recY[mby * 16:mby * 16 + 16, mbx * 16:mbx * 16 + 16] = rec
        recU[mby * mch:mby * mch + mch, mbx * mcw:mbx * mcw + mcw] = rc["recon"][2 * i][:mch, :mcw]
        recV[mby * mch:mby * mch + mch, mbx * mcw:mbx * mcw + mcw] = rc["recon"][2 * i + 1][:mch, :mcw]
    return {"luma": ry, "chroma": rc, "cbp": cbp, "cbp_blk": cbp_blk, "recon": (recY, recU, recV), "jobs_y": jobs_y, "jobs_c": jobs_c}


def deblock_frame(Y, U, V, yuv_format, mbs, blks, mvlimit=4):
    """jmo_deblock_frame (loopFilter.c:87) on 8-bit planes; mbs / blks use the ABI's dtypes (same layout as jmo_deblock_mb / _blk)."""
    L = lib()
    L.jmo_deblock_frame.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int]
    L.jmo_deblock_frame.restype = None
    H, W = Y.shape
    planes = [np.ascontiguousarray(p, np.uint16) if p is not None else None for p in (Y, U, V)]
    mbs = np.ascontiguousarray(mbs)
    blks = np.ascontiguousarray(blks)
    assert mbs.dtype.itemsize == 12 and blks.dtype.itemsize == 24
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    L.jmo_deblock_frame(ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), W, H, yuv_format, 8, ptr(mbs), ptr(blks), mvlimit)
    return [p.astype(np.uint8) if p is not None else None for p in planes]


# ------------------------------------------------------------------ EPZS / UMHexagonS state + the low-complexity P-slice driver

MAX_LIST, MAX_REFS, LC_REFS = 33, 32, 5


class EpzsConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("search_range", "bipred_me", "bipred_search_range", "pattern", "dual", "fixed", "temporal", "spatial_mem",
                                       "min_scale", "med_scale", "max_scale", "subpel_scale", "width", "height", "width_cr", "height_cr",
                                       "bitdepth_luma", "bitdepth_chroma", "chroma_me", "chroma_me_weight", "max_refs")]


class EpzsSlice(C.Structure):
    _fields_ = [("is_b_slice", C.c_int), ("poc", C.c_int), ("list_size", C.c_int * 2), ("list_poc", (C.c_int * MAX_LIST) * 2),
                ("num_ref_idx_l0_active", C.c_int), ("ref_pic_num_l0", C.c_longlong * MAX_LIST), ("col_mv", C.c_void_p * 2), ("col_ref_id", C.c_void_p * 2)]


class UmhexConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("search_range", "bipred_search_range", "dsr", "scale", "qp_n", "bipred_me", "full_search", "successive_bframe",
                                       "width", "height", "max_refs")]


class LowcplxParams(C.Structure):
    _fields_ = [("search_mode", C.c_int), ("search_range", C.c_int), ("num_refs", C.c_int), ("full_search", C.c_int), ("valid", C.c_int * 8),
                ("lambda_mf", C.c_int * 3), ("ref_cost1", C.c_int), ("md_metric", C.c_int), ("wp_pred", C.c_int),
                ("wp_weight", C.c_int * MAX_REFS), ("wp_offset", C.c_int * MAX_REFS), ("me", MeParams), ("epzs_subpel_me", C.c_int),
                ("W", C.c_int), ("H", C.c_int), ("slice_id", C.c_void_p), ("epzs", C.c_void_p), ("umhex", C.c_void_p),
                ("frame_ctr_b", C.c_int), ("img_number", C.c_int), ("blocktype_lut", (C.c_int * 4) * 4), ("all_mv_state", C.c_void_p),
                ("transform8x8_mode", C.c_int), ("q8", C.c_void_p)]


MB_INTER_DTYPE = np.dtype([("best_mode", "<i4"), ("min_cost", "<i4"), ("b8mode", "<i4", (4,)), ("b8ref", "<i4", (4,)),
                           ("final_mv", "<i2", (16, 2)), ("skip_mv", "<i2", (2,)),
                           ("pred", "<i2", (LC_REFS, 41, 2)), ("mv_int", "<i2", (LC_REFS, 41, 2)), ("mv", "<i2", (LC_REFS, 41, 2)),
                           ("cost_int", "<i4", (LC_REFS, 41)), ("cost", "<i4", (LC_REFS, 41)),
                           ("pred8ts", "<i2", (LC_REFS, 4, 2)), ("mv_int8ts", "<i2", (LC_REFS, 4, 2)), ("mv8ts", "<i2", (LC_REFS, 4, 2)),
                           ("cost_int8ts", "<i4", (LC_REFS, 4)), ("cost8ts", "<i4", (LC_REFS, 4)),
                           ("transform8x8_flag", "<i4"), ("cbp8ts", "<i4"), ("p8mode", "<i4", (4,)), ("p8ref", "<i4", (4,))], align=True)

# JM's shipped defaults (bin/encoder_*.cfg: EPZSPattern 2, Dual 3, Fixed 2, Temporal 1, SpatialMem 1, thresholds 0/1/2, sub-pel 2; UMHexDSR 1, UMHexScale 3)
EPZS_DEFAULTS = dict(pattern=2, dual=3, fixed=2, temporal=1, spatial_mem=1, min_scale=0, med_scale=1, max_scale=2, subpel_scale=2)


def _walker_protos():
    L = lib()
    if getattr(L, "_walker_ready", False):
        return L
    vp, ip = C.c_void_p, C.c_int
    L.jmo_epzs_create.restype = vp
    L.jmo_epzs_create.argtypes = [C.POINTER(EpzsConfig)]
    L.jmo_epzs_destroy.argtypes = [vp]
    L.jmo_epzs_slice_init.argtypes = [vp, C.POINTER(EpzsSlice)]
    L.jmo_epzs_search_count.restype = C.c_uint
    L.jmo_epzs_search_count.argtypes = [vp]
    L.jmo_epzs_alias_events.restype = C.c_long
    L.jmo_epzs_alias_events.argtypes = [vp]
    L.jmo_epzs_map_set.argtypes = [vp, vp, C.c_int]
    L.jmo_epzs_ideal_map.argtypes = [vp, C.c_int]
    L.jmo_epzs_first_touch.argtypes = [vp, vp]
    L.jmo_umhex_create.restype = vp
    L.jmo_umhex_create.argtypes = [C.POINTER(UmhexConfig)]
    L.jmo_umhex_destroy.argtypes = [vp]
    L.jmo_lowcplx_p_slice.argtypes = [C.POINTER(LowcplxParams), vp, vp, ip, vp, vp, ip, ip, vp]
    L.jmo_lowcplx_p_slice.restype = None
    L._walker_ready = True
    return L


class Epzs:
    def __init__(self, W, H, search_range, max_refs, yuv_format=1, **kw):
        c = EpzsConfig()
        o = dict(EPZS_DEFAULTS)
        o.update(kw)
        for k, v in o.items():
            setattr(c, k, v)
        c.search_range, c.width, c.height, c.max_refs = search_range, W, H, max_refs
        c.width_cr, c.height_cr = (W // 2, H // 2) if yuv_format == 1 else (W // 2, H) if yuv_format == 2 else (W, H)
        c.bitdepth_luma = c.bitdepth_chroma = 8
        self.cfg = c
        self.h = _walker_protos().jmo_epzs_create(C.byref(c))

    def slice_init(self, poc, list_pocs, ref_pic_nums, col_mv, col_ref_id, num_ref_idx_l0_active=None, is_b=False):
        """col_mv: two (H/4, W/4, 2) int16 arrays, col_ref_id: two (H/4, W/4) int64 arrays (the co-located pictures listX[list][0], [1])."""
        s = EpzsSlice()
        s.is_b_slice, s.poc = int(is_b), poc
        s.list_size[0] = len(list_pocs)
        for i, p in enumerate(list_pocs):
            s.list_poc[0][i] = p
            s.ref_pic_num_l0[i] = int(ref_pic_nums[i])
        s.num_ref_idx_l0_active = num_ref_idx_l0_active or len(list_pocs)
        self._keep = [np.ascontiguousarray(a, dtype=np.int16) for a in col_mv] + [np.ascontiguousarray(a, dtype=np.int64) for a in col_ref_id]
        s.col_mv[0], s.col_mv[1] = self._keep[0].ctypes.data, self._keep[1].ctypes.data
        s.col_ref_id[0], s.col_ref_id[1] = self._keep[2].ctypes.data, self._keep[3].ctypes.data
        lib().jmo_epzs_slice_init(self.h, C.byref(s))

    def search_count(self):
        """integer searches since the object was created (EPZSBlkCount without its 16-bit wrap)"""
        return int(_walker_protos().jmo_epzs_search_count(self.h))

    def map_set(self, stamps, blk_count):
        """EPZSMap ((2R+1, 2R+1) int16 or None = zeros) and EPZSBlkCount of a running encoder"""
        m = None if stamps is None else np.ascontiguousarray(stamps, dtype=np.int16)
        _walker_protos().jmo_epzs_map_set(self.h, None if m is None else m.ctypes.data, int(blk_count))

    def ideal_map(self, on=True):
        """NOT JM: a what-if that answers aliased tests as a map cleared per search would (shows that a test clip's aliases matter)"""
        _walker_protos().jmo_epzs_ideal_map(self.h, int(on))

    def first_touch(self):
        """(2R+1, 2R+1) uint32: ordinal of the first search that tested or stamped each map cell since map_set / creation, 0 = none"""
        side = 2 * self.cfg.search_range + 1
        out = np.zeros((side, side), np.uint32)
        _walker_protos().jmo_epzs_first_touch(self.h, out.ctypes.data)
        return out

    def alias_events(self):
        """map tests answered "visited" from a stamp 65536 k searches old, or from the initial zero (jmo_epzs.c map_shadow)"""
        return int(_walker_protos().jmo_epzs_alias_events(self.h))

    def close(self):
        if self.h:
            lib().jmo_epzs_destroy(self.h)
            self.h = None


class Umhex:
    def __init__(self, W, H, search_range, max_refs, qp_n, dsr=1, scale=3, full_search=2, successive_bframe=0):
        c = UmhexConfig()
        c.search_range, c.width, c.height, c.max_refs, c.qp_n = search_range, W, H, max_refs, qp_n
        c.dsr, c.scale, c.full_search, c.successive_bframe = dsr, scale, full_search, successive_bframe
        self.cfg = c
        self.h = _walker_protos().jmo_umhex_create(C.byref(c))

    def close(self):
        if self.h:
            lib().jmo_umhex_destroy(self.h)
            self.h = None


BLOCKTYPE_LUT = {(0, 0): 7, (0, 1): 6, (1, 0): 5, (1, 1): 4, (1, 3): 3, (3, 1): 2, (3, 3): 1}     # configfile.c:830-836


def lowcplx_params(search_mode, search_range, num_refs, lambda_mf, ref_cost1, W, H, epzs=None, umhex=None, full_search=2, metric=(0, 2, 2),
                   md_metric=2, valid=(1, 1, 1, 1, 1, 1, 1), frame_ctr_b=0, img_number=1, level_mv=(-511, 511), all_mv_state=None,
                   transform8x8_mode=0, qp=28, cavlc=1, rdopt=0):
    q = LowcplxParams()
    q.search_mode, q.search_range, q.num_refs, q.full_search = search_mode, search_range, num_refs, full_search
    for m in range(1, 8):
        q.valid[m] = valid[m - 1]
    q.lambda_mf[0], q.lambda_mf[1], q.lambda_mf[2] = lambda_mf
    q.ref_cost1, q.md_metric = ref_cost1, md_metric
    q.me = me_params(rdopt=rdopt, metric=metric, level_mv=level_mv, transform8x8_mode=transform8x8_mode)      # rdopt != 0: what BlockMotionSearch does in the high-complexity modes; the decision stays the low-complexity one
    q.transform8x8_mode = transform8x8_mode
    if transform8x8_mode:                    # the inter 8x8 luma quantiser of the slice: flat matrices, default offset (342 / 2048), no adaptive rounding
        ls, ils, lo, _ = flat_tables(qp, 342, True)
        q._keep_q8 = QuantHolder(dict(levelscale=ls, invlevelscale=ils, leveloffset=lo, qp=qp, adaptive_rounding=0, adapt_rnd_weight=0, field_scan=0,
                                      disthres=0, max_val=255, cavlc=cavlc, img_qp=qp, transform8x8_flag=1))
        q.q8 = C.addressof(q._keep_q8.c)
    q.epzs_subpel_me = 1
    q.W, q.H = W, H
    q.epzs = epzs.h if epzs else None
    q.umhex = umhex.h if umhex else None
    q.frame_ctr_b, q.img_number = frame_ctr_b, img_number
    for (a, b), v in BLOCKTYPE_LUT.items():
        q.blocktype_lut[a][b] = v
    if all_mv_state is not None:             # np.int16 (4, 4, MAX_REFS, 9, 2), carried from call to call by the caller
        q._keep_all_mv = all_mv_state
        q.all_mv_state = all_mv_state.ctypes.data
    return q


def lowcplx_p_slice(q, refpics, curY, ref_idx=None, mv=None, mb_first=0, mb_count=None):
    """Runs macroblocks [mb_first, mb_first+mb_count) of a P picture. Returns (records, ref_idx, mv): per-macroblock MB_INTER_DTYPE records and the
    picture-level LIST_0 arrays ((H/4, W/4) int8, (H/4, W/4, 2) int16), updated in place when passed in."""
    L = _walker_protos()
    W, H = q.W, q.H
    if ref_idx is None:
        ref_idx = np.full((H // 4, W // 4), -1, np.int8)
        mv = np.zeros((H // 4, W // 4, 2), np.int16)
    n = (W // 16) * (H // 16) if mb_count is None else mb_count
    out = np.zeros(n, MB_INTER_DTYPE)
    refs = (Ref * len(refpics))(*[r.ref for r in refpics])
    cur = np.ascontiguousarray(curY, dtype=np.uint16)
    L.jmo_lowcplx_p_slice(C.byref(q), refs, cur.ctypes.data, cur.shape[1], ref_idx.ctypes.data, mv.ctypes.data, mb_first, n, out.ctypes.data)
    return out, ref_idx, mv


def epzs_colocated(epzs, W, H):
    """EPZSCo_located->mv[LIST_0] of the oracle's state after slice_init: (H/4, W/4, 2) int16."""
    L = _walker_protos()
    L.jmo_epzs_colocated.restype = C.c_void_p
    L.jmo_epzs_colocated.argtypes = [C.c_void_p]
    p = L.jmo_epzs_colocated(epzs.h)
    n = (H // 4) * (W // 4) * 2
    return np.ctypeslib.as_array((C.c_short * n).from_address(p)).reshape(H // 4, W // 4, 2).copy()


def umhex_thresholds(umhex):
    L = _walker_protos()
    ia, fa = C.c_int * 8, C.c_float * 8
    a = [ia(), ia(), ia(), ia(), fa(), fa(), fa()]
    L.jmo_umhex_thresholds.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    L.jmo_umhex_thresholds(umhex.h, *a)
    return [list(x) for x in a]


def epzs_threshold(epzs, which, bt):
    L = _walker_protos()
    L.jmo_epzs_threshold.argtypes = [C.c_void_p, C.c_int, C.c_int]
    return L.jmo_epzs_threshold(epzs.h, which, bt)
