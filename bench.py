#!/usr/bin/env python3
"""bench.py -- macroblocks/sec of the JM per-macroblock hot path on MI355X.

One step = one pass of the hot path over one 1080p P-frame (BASELINE.json configs[1]: 1920x1080 coded as
1920x1088 = 8160 macroblocks, 4:2:0, baseline tools, FullSearch +-32, one reference):
    getSubImagesLuma + getSubImagesChroma of the reference        (jmhip_interp_luma / _chroma)
    integer full search, all 41 partitions, SAD + MV cost          (jmhip_me_frame: me_int kernel)
    half/quarter-pel refinement with SATD                          (me_sub kernel)
    MC prediction, residual, dct_4x4 x16 + dct_chroma x2, coefficient-cost thresholds, reconstruction
                                                                   (jmhip_residual_frame)
    recon -> next reference                                        (D2D copy; for N > 1 the per-frame RCCL all-gather)
Source frames, reference and per-macroblock jobs are resident in HBM before the timed region.

N > 1 (one process per GPU under torchrun): the frame is cut into N slices of whole macroblock rows
(SliceMode=1 style), every rank interpolates the full reference locally, searches/transforms its own slice, and the
reconstructed bands are all-gathered over RCCL once per frame (strong scaling: total work fixed).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HIP-event timed on the
library's stream) and `cpu_baseline` (the oracle port on one host core, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H_SRC, H, R = 1920, 1080, 1088, 32
MBW, MBH = W // 16, H // 16
QP = 28
ME_BYTES_PER_MB = 1352          # cur 512 + ref 512 + out 41*8 (u16 pels as in JM)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
# Integer-VALU model of the search kernel (the bound that actually binds): per candidate and lane the algorithm
# needs 64 v_sad_u8-class ops (16 rows x 4 dwords), 25 adds for the SetupLargerBlocks tree, and per partition one add
# (mv cost) and half a v_min3_u32 (one min3 folds the keys of two candidates). Issue times per wave-instruction per SIMD measured
# on MI355X with tools/ubench/valu_rate2.hip (profiles/r01_valu_issue_rates.txt): v_sad_u8 / v_min3_u32 1.89 ns, v_add_u32 1.04 ns.
# 1024 SIMDs. (Until the min3 pairing was built the model charged one v_min_u32 per partition: 266 ns.)
VALU_NS_PER_WAVE_CANDIDATE = 64 * 1.89 + 25 * 1.04 + 40 * 1.04 + 20.5 * 1.89
N_SIMD = 1024
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_final_summary.json")


def pmc_traffic(kernel, n_units):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/profile.sh: FETCH_SIZE and
    WRITE_SIZE in separate runs, KB units). Only meaningful for the full-frame single-GPU launch it was recorded on."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
            k = d.get(kernel) or d[kernel + "<false>"]        # (the kernel is a template since round 3: <false> = the frame form, <true> = device-resident item lists)
        if n_units != MBW * MBH:
            return None
        return int((k["FETCH_SIZE_KB_raw_per_launch"] + k["WRITE_SIZE_KB_raw_per_launch"]) * 1024)
    except Exception:
        return None


def synth_frames(nframes, noise_clip=False):
    """SURVEY 8(d) recipe: blurred random field translated by (+4,-4)/frame + N(0,2) noise; chroma from luma.
    noise_clip: the recipe's adversarial second clip, i.i.d. uniform samples (every partition finds its own vector)."""
    if noise_clip:
        frames = []
        for f in range(nframes):
            r = np.random.default_rng(1000 + f)
            Y = r.integers(0, 256, (H, W), dtype=np.uint8)
            frames.append((Y, r.integers(0, 256, (H // 2, W // 2), dtype=np.uint8), r.integers(0, 256, (H // 2, W // 2), dtype=np.uint8)))
        return frames
    rng = np.random.default_rng(20260410)
    B = rng.integers(0, 256, (H // 8 + 16, W // 8 + 16)).astype(np.float64)
    B = np.kron(B, np.ones((8, 8)))
    k = np.ones(9) / 9.0
    B = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, B)
    B = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, B)
    frames = []
    for f in range(nframes):
        dx, dy = 5 + 4 * (f - 1) if f else 0, -3 - 4 * (f - 1) if f else 0
        noise = np.random.default_rng(f + 1).normal(0, 2.0, (H, W))
        Y = np.clip(np.round(B[40 + dy:40 + dy + H, 40 + dx:40 + dx + W] + noise), 0, 255).astype(np.uint8)
        Y[H_SRC:] = Y[H_SRC - 1]                      # JM pads 1080 -> 1088 by replication
        sub = Y[::2, ::2].astype(np.float64)
        U = np.clip(np.round(128 + 0.25 * (sub - 128)), 0, 255).astype(np.uint8)
        V = np.clip(np.round(128 - 0.25 * (sub - 128)), 0, 255).astype(np.uint8)
        frames.append((Y, U, V))
    return frames


def lambda_factor(qp):
    return int(65536 * np.sqrt(0.85 * 2 ** ((qp - 12) / 3.0)) + 0.5)     # slice.c:1329-1358 (P slice, rdopt on)


def make_jobs(pkg, rows):
    """Per-macroblock predictor field (16,-16) + U{-8..8} quarter-pel (SURVEY 8(d) 'independent MBs' run)."""
    rng = np.random.default_rng(7)
    # true motion of the clip is (+4,-4) pel per frame: a predictor near it, as JM's median predictor would be
    pred = rng.integers(-8, 9, (MBH, MBW, 2)) + np.array([16, -16])
    mbs = np.zeros(len(rows) * MBW, dtype=pkg.ME_MB_DTYPE)
    i = 0
    for y in rows:
        for x in range(MBW):
            mbs[i]["mb_x"], mbs[i]["mb_y"], mbs[i]["ref"], mbs[i]["ref_is_0"] = x, y, 0, 1
            mbs[i]["pred_mv"][:] = pred[y, x]
            i += 1
    return mbs


JM_CFG = """
InputFile = "synth1080.yuv"
InputHeaderLength = 0
StartFrame = 0
FramesToBeEncoded = 2
FrameRate = 30.0
SourceWidth = 1920
SourceHeight = 1080
OutputFile = "out.264"
ReconFile = "out_rec.yuv"
TraceFile = "trace_enc.txt"
ProfileIDC = 66
LevelIDC = 40
IntraPeriod = 0
QPISlice = %d
QPPSlice = %d
SearchRange = %d
NumberReferenceFrames = 1
NumberBFrames = 0
SymbolMode = 0
SearchMode = -1
RDOptimization = 1
MEDistortionFPel = 0
MEDistortionHPel = 2
MEDistortionQPel = 2
MDDistortion = 2
ChromaMCBuffer = 1
ChromaMEEnable = 0
RestrictSearchRange = 2
AdaptiveRounding = 1
Transform8x8Mode = 0
LoopFilterDisable = 0
"""


def cpu_baseline_reference(frames):
    """The REAL JM encoder (oracle/_ref/jm_plain, compiled from the reference sources in the build container; the
    binary travels, the sources do not) on the first two frames of the same synthetic clip: I + P, FullSearch +-32,
    1 reference, baseline tools. JM reports its own per-frame total and motion-estimation times (image.c:381,673;
    mv-search.c:597-603). Single-threaded, like JM always is."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "jm_plain")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "synth1080.yuv"), "wb") as f:
            for (Y, U, V) in frames[:2]:
                f.write(Y[:H_SRC].tobytes()); f.write(U[:H_SRC // 2].tobytes()); f.write(V[:H_SRC // 2].tobytes())
        with open(os.path.join(d, "min.cfg"), "w") as f:
            f.write(JM_CFG % (QP, QP, R))
        t0 = time.perf_counter()
        try:
            r = subprocess.run([exe, "-d", "min.cfg"], cwd=d, capture_output=True, text=True, timeout=300)
        except Exception:
            return None
        wall = time.perf_counter() - t0
        m = re.search(r"^0001\(P\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", r.stdout, re.M)
        if not m:          # JM's main() returns a nonzero status even on success
            return None
        tot_ms, me_ms = int(m.group(1)), int(m.group(2))
    return {"value": round(MBW * MBH / (me_ms * 1e-3), 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference",
            "sample": "real JM 12.4 lencod (oracle/_ref), frames 0-1 of the same clip (I+P), P-frame of %d MBs: motion estimation %d ms "
                      "(value = MBs / ME time, the figure most favourable to the CPU), whole P-frame %d ms = %.1f MB/s incl. mode decision and "
                      "CAVLC; run wall %.1f s; host has %d cores, 1 used (JM is single-threaded)"
                      % (MBW * MBH, me_ms, tot_ms, MBW * MBH / (tot_ms * 1e-3), wall, os.cpu_count())}


def jm_end_to_end(frames, config3=False, rdopt1=False):
    """The reference encoder itself, same 1080p clip (frames 0-1: I + P), FullSearch +-32, low-complexity decision with intra off in the P
    picture (RDOptimization 0, DisableIntraInInter 1 -- the configuration whose whole P-slice search + inter decision is ONE device call):
    the unmodified JM (oracle/_ref/jm_plain) against JM bound to libjmhip.so at slice level (oracle/_ref/jm_hip, integration/jm_shim.c,
    mask 0xd801 = sub-pel planes (fetched into JM's rows only when a forwarded call is about to read them) + slice binding + the slice's frame stage (4:2:0, 4x4 transform: JM's prediction and dct_4x4 / dct_chroma calls
    answered from the device's records) + loop filter on the device, everything else JM's own code). Wall-clock of whole encodes."""
    import re
    import subprocess
    import tempfile
    exes = [os.path.join(ROOT, "oracle", "_ref", e) for e in ("jm_plain", "jm_hip")]
    if not all(os.path.exists(e) for e in exes):
        return None
    out = {}
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "synth1080.yuv"), "wb") as f:
            for (Y, U, V) in frames[:2]:
                f.write(Y[:H_SRC].tobytes()); f.write(U[:H_SRC // 2].tobytes()); f.write(V[:H_SRC // 2].tobytes())
        cfg = (JM_CFG % (QP, QP, R)).replace("RDOptimization = 1", "RDOptimization = 0") + "DisableIntraInInter = 1\n"
        if rdopt1:                                       # JM's default kind of configuration: rate-distortion optimised decision, intra candidates in P pictures
            cfg = JM_CFG % (QP, QP, R)
        if config3:                                      # BASELINE config 3's tools: EPZS, Hadamard SAD at every level, 8x8 transform enabled, CABAC
            for a, b in (("ProfileIDC = 66", "ProfileIDC = 100"), ("SymbolMode = 0", "SymbolMode = 1"), ("SearchMode = -1", "SearchMode = 3"),
                         ("MEDistortionFPel = 0", "MEDistortionFPel = 2"), ("Transform8x8Mode = 0", "Transform8x8Mode = 1"), ("AdaptiveRounding = 1", "AdaptiveRounding = 0")):
                assert a in cfg
                cfg = cfg.replace(a, b)
        with open(os.path.join(d, "min.cfg"), "w") as f:
            f.write(cfg)
        digests = []
        for exe, key in zip(exes, ("jm_plain", "jm_hip")):
            env = dict(os.environ, JMHIP_SHIM="b801" if rdopt1 else "d801", JMHIP_SHIM_STATS="1")
            t0 = time.perf_counter()
            try:
                r = subprocess.run([exe, "-d", "min.cfg"], cwd=d, env=env, capture_output=True, text=True, timeout=400)
            except Exception:
                return None
            out[key + "_s"] = round(time.perf_counter() - t0, 2)
            if os.environ.get("JMHIP_E2E_STDERR"):           # diagnostics (e.g. with JMHIP_SLICE_TRACE=1): keep what the encoders wrote to stderr
                with open(os.environ["JMHIP_E2E_STDERR"], "a") as f:
                    f.write("==== %s\n%s\n" % (key, r.stderr))
            m = re.search(r"^0001\(P\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", r.stdout, re.M)
            if not m:
                sys.stderr.write("jm_end_to_end: %s did not report a P frame (exit %d): %s\n" % (key, r.returncode, (r.stderr or r.stdout)[-600:]))
                return None
            out[key + "_p_frame_ms"] = int(m.group(1))
            out[key + "_p_frame_me_ms"] = int(m.group(2))
            with open(os.path.join(d, "out.264"), "rb") as f:
                digests.append(f.read())
            if key == "jm_hip":
                mm = re.search(r"^\s*BlockMotionSearch\s+device\s+(\d+)\s+forwarded\s+(\d+)", r.stderr, re.M)
                out["block_motion_search_calls_served"] = int(mm.group(1)) if mm else None
                out["block_motion_search_calls_forwarded"] = int(mm.group(2)) if mm else None
                # wall time the shim spent inside its coarse device-side hooks, whole encode (transfers and layout conversion included)
                out["jm_hip_hooks"] = {k.strip(): {"calls": int(n), "ms": float(v), "last_call_ms": float(l)} for k, n, v, l in
                                       re.findall(r"^\s*(\S[^\n]*?)\s+device\s+(\d+)\s+forwarded\s+\d+\s+([\d.]+) ms inside the hook \(last call ([\d.]+) ms\)", r.stderr, re.M)}
                for sym, key2 in (("dct_4x4 (slice records)", "dct_4x4_calls"), ("dct_chroma (slice records)", "dct_chroma_calls"), ("LumaPrediction (slice)", "luma_prediction_calls"),
                                  ("ChromaPrediction4x4 (slice)", "chroma_prediction_calls")):
                    mm = re.search(r"^\s*%s\s+device\s+(\d+)\s+forwarded\s+(\d+)" % re.escape(sym), r.stderr, re.M)
                    if mm:
                        out[key2 + "_served_from_slice_records"] = int(mm.group(1)); out[key2 + "_left_to_jm_in_bound_slices"] = int(mm.group(2))
                ms = re.search(r"slice binding: (\d+) slices, (\d+) kernel passes", r.stderr)
                if ms:
                    out["jm_hip_slice_sweeps"] = int(ms.group(2))
        out["bitstreams_identical"] = digests[0] == digests[1]
        if not rdopt1 and not config3 and len(frames) >= 4:
            # the bound encoder alone over four frames (I P P P): what a P frame costs once the one-time work of the first one -- device
            # allocations of the slice search (1 GB of SAD surfaces), first-touch pages of the record buffer -- is behind it
            with open(os.path.join(d, "synth1080.yuv"), "wb") as f:
                for (Y, U, V) in frames[:4]:
                    f.write(Y[:H_SRC].tobytes()); f.write(U[:H_SRC // 2].tobytes()); f.write(V[:H_SRC // 2].tobytes())
            with open(os.path.join(d, "min4.cfg"), "w") as f:
                f.write(cfg.replace("FramesToBeEncoded = 2", "FramesToBeEncoded = 4"))
            try:
                r = subprocess.run([exes[1], "-d", "min4.cfg"], cwd=d, env=dict(os.environ, JMHIP_SHIM="d801", JMHIP_SHIM_STATS="1"),
                                   capture_output=True, text=True, timeout=200)
                pf = re.findall(r"^000\d\(P\)\s+\d+\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(\d+)\s+(\d+)", r.stdout, re.M)
                mh = re.search(r"P slices \(one device call each\)\s+device\s+\d+\s+forwarded\s+\d+\s+[\d.]+ ms inside the hook \(last call ([\d.]+) ms\)", r.stderr)
                if len(pf) == 3:
                    out["jm_hip_four_frames"] = {"p_frame_ms": [int(a) for a, _ in pf], "slice_hook_last_call_ms": float(mh.group(1)) if mh else None}
            except Exception:
                pass
        if rdopt1:
            out["config"] = ("1920x1080 I+P, FullSearch +-32, 1 reference, RDOptimization 1, intra candidates in the P picture, CAVLC; jm_hip: JMHIP_SHIM=0xb801 "
                             "(speculative slice binding: a BlockMotionSearch call is answered from the device's record when JM's predictor equals the recorded one, "
                             "else JM's own search runs)")
            return out
        out["config"] = ("1920x1080 I+P, EPZS +-32, Hadamard SAD at every level, Transform8x8Mode 1, CABAC, 1 reference, RDOptimization 0, DisableIntraInInter 1; jm_hip: JMHIP_SHIM=0xd801 (the 8x8 transform keeps the frame stage in JM)"
                         if config3 else "1920x1080 I+P, FullSearch +-32, 1 reference, RDOptimization 0, DisableIntraInInter 1, CAVLC; jm_hip: JMHIP_SHIM=0xd801 (slice search + frame stage bound at slice level, sub-pel planes left on the device)")
    return out


def cpu_baseline(pkg, frames, n_mbs):
    """The oracle's restated JM algorithm on ONE host core over a bounded sample of the same workload."""
    from tests import oracle
    L = oracle.lib()
    (Y1, U1, V1), (Y0, U0, V0) = frames[1], frames[0]
    t0 = time.perf_counter()
    rp = oracle.RefPic(Y0, U0, V0, yuv_format=1)      # getSubImagesLuma/Chroma on the CPU
    t_interp = time.perf_counter() - t0
    p = oracle.me_params(rdopt=1)
    rows = list(range(MBH // 2, MBH // 2 + max(1, n_mbs // MBW)))
    mbs = make_jobs(pkg, rows)
    n = len(mbs)
    xy = np.ascontiguousarray(np.stack([mbs["mb_x"], mbs["mb_y"]], 1).astype(np.int16))
    preds = np.ascontiguousarray(mbs["pred_mv"].astype(np.int16))
    lam = (ctypes.c_int * 3)(*[lambda_factor(QP)] * 3)
    ql = oracle.QuantHolder(pkg.flat_quant(QP, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1))
    qc = oracle.QuantHolder(pkg.flat_quant(QP, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1))
    cur = [np.ascontiguousarray(a, dtype=np.uint16) for a in (Y1, U1, V1)]
    vp = ctypes.c_void_p
    L.jmo_hotpath_mbs.restype = ctypes.c_longlong
    L.jmo_hotpath_mbs.argtypes = [ctypes.POINTER(oracle.MeParams), ctypes.POINTER(oracle.Ref), vp, vp, vp, ctypes.c_int, vp, vp, ctypes.c_int,
                                  ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(oracle.Quant), ctypes.POINTER(oracle.Quant), vp, vp]
    t0 = time.perf_counter()
    L.jmo_hotpath_mbs(ctypes.byref(p), ctypes.byref(rp.ref), cur[0].ctypes.data, cur[1].ctypes.data, cur[2].ctypes.data, W,
                      xy.ctypes.data, preds.ctypes.data, n, R, lam, ctypes.byref(ql.c), ctypes.byref(qc.c), None, None)
    t_mb = time.perf_counter() - t0
    per_mb = t_mb / n + t_interp / (MBW * MBH)         # interpolation is per frame: charge its per-MB share
    return {"value": round(1.0 / per_mb, 1), "unit": "macroblocks/s", "cores": 1, "kind": "port",
            "sample": "%d MBs (rows %d..%d of frame 1, FullSearch +-%d, 41 partitions + sub-pel + TQ) in %.1f s; "
                      "+ full-frame sub-pel plane generation %.2f s charged per MB; host has %d cores, 1 used (JM is single-threaded)"
                      % (n, rows[0], rows[-1], R, t_mb, t_interp, os.cpu_count())}


def parity_check(pkg, ctx, frames, mbs, prm, last_src, prev_ref):
    """A sample of the LAST step's motion search (integer + sub-pel vectors and costs, all 41 partitions) against the oracle's restatement of JM on
    the same inputs: the source frame of that step and the reference picture it searched (downloaded before the step)."""
    from tests import oracle
    if prev_ref is None:
        return None
    res = ctx.me_results(len(mbs))
    rng = np.random.default_rng(99)
    pick = np.sort(rng.choice(len(mbs), size=12, replace=False))
    rp = oracle.RefPic(prev_ref, yuv_format=0)
    lam = [int(prm.lambda_[0]), int(prm.lambda_[1]), int(prm.lambda_[2])]
    want = oracle.me_frame(oracle.me_params(rdopt=1), [rp], frames[last_src][0], mbs[pick], -1, R, lam)
    ok = all(np.array_equal(res[k][pick], want[k]) for k in ("mv_int", "cost_int", "mv", "cost"))
    return {"macroblocks": int(len(pick)), "partitions_each": 41, "fields": ["mv_int", "cost_int", "mv", "cost"], "vs": "oracle (CPU restatement of JM)", "ok": bool(ok)}


def slice_search_times(pkg, ctx, lam, src, order):
    """For information: the JM-exact form of the search -- jmhip_p_slice_search: predictors, search, sub-pel, skip shortcut and the low-complexity
    inter decision on the device with JM's raster-order dependencies -- on the bench clip, one reference, per search mode. Two NEW pictures per
    mode: the first starts from an empty state, the second from what the first left (the relaxation schedule's first guess, as in a sequence);
    `sweeps` = relaxation sweeps until nothing changed."""
    import ctypes
    lib = pkg.load_library()
    out = {}
    ctx.interp_luma(0)                                  # the last step left a new integer picture in the slot
    # the last entry is BASELINE config 3's search: EPZS, Hadamard SAD at every level, both transform sizes competing (Transform8x8Mode 1)
    for name, mode, metric, t8 in (("FullSearch", -1, (0, 2, 2), 0), ("FastFullSearch", 0, (0, 2, 2), 0), ("EPZS", 3, (0, 2, 2), 0), ("UMHexagonS", 1, (0, 2, 2), 0), ("UMHexagonS_simplified", 2, (0, 2, 2), 0),
                                   ("EPZS_satd_transform8x8", 3, (2, 2, 2), 1)):
        ctx.slice_state_reset()
        ctx.epzs_colocated_upload(np.zeros((H // 4, W // 4, 2), np.int16))
        p = pkg.slice_host.slice_params(mode, R, 1, [lam] * 3, 10, W, H=H, metric=metric, t8=t8, qp_n=QP)
        lib.jmhip_epzs_scales(p, 2, (ctypes.c_int * 1)(0), 1)
        ts, sw = [], []
        for k in order[:2]:
            Y, U, V = src[k]
            ctx.cur_bind(Y.data_ptr(), U.data_ptr(), V.data_ptr())
            ctx.sync()
            t0 = time.perf_counter()
            ctx.p_slice_search(p, download=False)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
            sw.append(ctx.slice_passes())
        out[name] = {"ms_first_picture": round(ts[0] * 1e3, 1), "ms_next_picture": round(ts[1] * 1e3, 1), "sweeps": sw,
                     "macroblocks_per_s": round(MBW * MBH / ts[1], 1)}
    Y, U, V = src[order[0]]                              # leave the picture that follows the resident reference bound
    ctx.cur_bind(Y.data_ptr(), U.data_ptr(), V.data_ptr())
    return out


def per_partition_predictors(pkg, ctx, lam, prm, mbs):
    """For information: the integer search fed with JM's OWN predictors -- one per partition, as SetMotionVectorPredictor yields them in raster order
    (taken from jmhip_p_slice_search, FullSearch, of the bench picture) -- instead of the metric's one predictor per macroblock. FullSearch centres
    its window on each partition's predictor (mv-search.c:752-762): a macroblock costs one window walk per DISTINCT centre."""
    import ctypes
    lib = pkg.load_library()
    ctx.slice_state_reset()
    p = pkg.slice_host.slice_params(-1, R, 1, [lam] * 3, 10, W, H=H)
    rec = ctx.p_slice_search(p)
    pp = mbs.copy()
    pp["pred_mv"] = rec["pred"][:, 0]
    centres = np.clip(np.trunc(pp["pred_mv"].astype(np.int32) / 4.0).astype(np.int32), -2047 + R, 2047 - R)      # rdopt = 1: no clamp to the range
    distinct = np.array([len({(int(c[0]), int(c[1])) for c in m}) for m in centres])
    ctx.timing_enable(True)
    ctx.timing_select(["me_int", "me_sub"])
    ctx.timing_read()
    ctx.me_frame_async(prm, pp)
    for _ in range(3):
        ctx.me_frame_async(prm, None, len(pp))
    ctx.sync()
    t = ctx.timing_read()
    ctx.timing_enable(False)
    ctx.me_frame_async(prm, mbs)                       # leave the metric's job set resident again
    ctx.sync()
    return {"predictors": "JM's own, per partition (jmhip_p_slice_search, FullSearch)", "distinct_centres_per_mb": {"mean": round(float(distinct.mean()), 2), "max": int(distinct.max())},
            "me_int_ms": round(t["me_int"][0] / max(1, t["me_int"][1]), 4), "me_sub_ms": round(t["me_sub"][0] / max(1, t["me_sub"][1]), 4),
            "note": "one window walk per distinct centre (the union-window kernel only takes partial partition masks and ranges the fast kernel does not cover)"}


def main():
    global W, H_SRC, H, MBW, MBH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-mbs", type=int, default=4 * MBW, help="macroblocks in the CPU baseline sample (0 = skip)")
    ap.add_argument("--clip", choices=["translation", "noise"], default="translation",
                    help="translation = SURVEY 8(d)'s clip (default, the metric's workload); noise = its adversarial i.i.d. clip, for information only")
    ap.add_argument("--chroma-planes", action="store_true",
                    help="materialise the 2 x 64 eighth-pel chroma planes per reference (getSubImagesChroma, JM's ChromaMCBuffer = 1) instead of "
                         "computing the chroma prediction samples in the frame stage; same results, +0.019 ms per 1080p frame")
    ap.add_argument("--deblock", action="store_true",
                    help="for information: also run the in-loop deblocking filter on the device each frame (jmhip_deblock_recon; idc 0 on one GPU, "
                         "idc 2 with one slice per rank on N GPUs). Not part of the metric; the filter is a serial wavefront (DESIGN.md section 3)")
    ap.add_argument("--deblock-slices", type=int, default=0,
                    help="with --deblock on ONE GPU: filter as if the picture were cut into this many row slices with idc 2 "
                         "(what an N-GPU run does), to compare reference checksums")
    ap.add_argument("--pred", choices=["one", "per-partition"], default="one",
                    help="one = one predictor per macroblock (the metric's workload, SURVEY 8(d)); per-partition = JM's own predictors, one per partition, "
                         "as SetMotionVectorPredictor yields them in raster order (from jmhip_p_slice_search of the first picture): FullSearch then "
                         "walks one window per DISTINCT centre of a macroblock (N = 1 only; for information)")
    ap.add_argument("--exact", action="store_true",
                    help="the JM-EXACT form of the step: instead of the independent-macroblocks search with given predictors, every rank runs "
                         "jmhip_p_slice_search on ITS slice (predictors, FullSearch +-32, sub-pel, skip shortcut and the low-complexity decision on the "
                         "device, raster-order dependencies kept: JM's vectors), hands the slice to the frame stage (jmhip_slice_to_frame_band -> "
                         "jmhip_residual_frame) and the bands are exchanged as in the default step. For information next to the metric's workload")
    ap.add_argument("--exact-slices", type=int, default=0,
                    help="with --exact on ONE GPU: cut the picture into this many slices of whole macroblock rows, all searched in one call (slice_mbs) -- "
                         "what N ranks compute between them; the reference checksum must equal the N-rank run's")
    ap.add_argument("--size", choices=["1080p", "2160p"], default=None,
                    help="1080p = BASELINE configs[1], the configuration the metric is quoted on: the default at --gpus 1; 2160p = BASELINE configs[3] "
                         "(4K, FullSearch +-32, slices sharded across the GPUs with the reference picture gathered once per frame), the configuration "
                         "BASELINE names for more than one GPU: the default at --gpus > 1 (north_star's scaling target is quoted on 4K)")
    args = ap.parse_args()
    if args.size is None:
        args.size = "1080p" if args.gpus == 1 else "2160p"
    if args.size == "2160p":
        W, H_SRC, H = 3840, 2160, 2160
        MBW, MBH = W // 16, H // 16
        args.cpu_mbs = 0                              # the CPU baseline is quoted on the metric's own workload only

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    import torch
    import __graft_entry__ as ge
    pkg = ge._load_pkg()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # JMHIP_BENCH_SOLO="r/w": run rank r of w ALONE (no process group; the all-gather replaced by a local copy of the own chunk) to time
    # one rank's share of an N-GPU step on a single GPU. A diagnostic: the line it prints is not a bench result.
    solo = os.environ.get("JMHIP_BENCH_SOLO")
    if solo:
        rank, world = (int(v) for v in solo.split("/"))
        args.gpus = world
        args.cpu_mbs = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    # JMHIP_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on ONE GPU (every rank on cuda:0, gloo instead of RCCL, the band
    # exchange staged through host memory). For testing the multi-process path where no multi-GPU node is at hand; the numbers mean nothing.
    rehearsal = os.environ.get("JMHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # JMHIP_BENCH_RCCL1=1: take the N > 1 code path (bands, one-chunk all-gather over RCCL on the context's stream, scatter) with a process
    # group of ONE rank: what a one-GPU box can exercise of the RCCL transport (tests/test_bench_ranks.py). Not a bench result.
    rccl1 = os.environ.get("JMHIP_BENCH_RCCL1") == "1" and world == 1 and not solo
    multi = world > 1 or rccl1
    if rccl1:
        args.cpu_mbs = 0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if multi and not solo:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)

    # ---- slice of this rank: whole macroblock rows, B = ceil(MBH / world) rows per rank
    from h264_amd import slices
    row0, row1, band = slices.band_rows(MBH, world, rank)
    rows = list(range(row0, row1))

    nframes = 4
    frames = synth_frames(nframes, args.clip == "noise")
    if args.clip == "noise":
        args.cpu_mbs = 0
    ctx = pkg.Context(W, H, yuv_format=1, max_refs=1, search_range=R, device=local_rank)
    src = [[torch.from_numpy(p).to(dev) for p in f] for f in frames]      # source frames resident in HBM
    ctx.ref_upload(0, *frames[0])
    mbs = make_jobs(pkg, rows)
    n = len(mbs)
    lam = lambda_factor(QP)
    prm = pkg.MeParams()
    prm.search_mode, prm.search_range, prm.rdopt, prm.is_b_slice = -1, R, 1, 0
    prm.level_mv_min, prm.level_mv_max = -511, 511
    prm.lambda_[0] = prm.lambda_[1] = prm.lambda_[2] = lam
    prm.transform8x8_mode, prm.subpel, prm.partition_mask = 0, 1, (1 << 41) - 1
    if args.exact:
        args.cpu_mbs = 0
        ctx.slice_state_reset()
        first_mb, count_mb = row0 * MBW, (row1 - row0) * MBW
        sp = pkg.slice_host.slice_params(-1, R, 1, [lam] * 3, 10, W, H=H, mb_first=first_mb, mb_count=max(count_mb, 1))
        if not multi and args.exact_slices > 1:
            sp.slice_mbs = -(-MBH // args.exact_slices) * MBW
    if args.pred == "per-partition":
        if multi:
            sys.exit("bench.py --pred per-partition runs on one GPU")
            Y0, U0, V0 = src[1]
        ctx.cur_bind(Y0.data_ptr(), U0.data_ptr(), V0.data_ptr())
        ctx.interp_luma(0)
        ctx.slice_state_reset()
        rec = ctx.p_slice_search(pkg.slice_host.slice_params(-1, R, 1, [lam] * 3, 10, W, H=H))
        mbs = mbs.copy()
        mbs["pred_mv"] = rec["pred"][:, 0]
        args.cpu_mbs = 0                                  # the extras below are the default workload's
    quants = np.array([pkg.flat_quant(QP, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1),
                       pkg.flat_quant(QP, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1),
                       pkg.flat_quant(QP + 3, 342, adaptive_rounding=1, adapt_rnd_weight=4, cavlc=1)], dtype=pkg.QUANT_DTYPE)
    # N > 1: one exchange buffer per rank -- its band as a single chunk [Y | U | V] -- and one gather buffer of `world` chunks
    chunk = ctx.band_chunk_bytes(band) if multi else 0
    sbuf = torch.zeros(max(chunk, 4), dtype=torch.uint8, device=dev)
    gbuf = torch.zeros(max(chunk * world, 4), dtype=torch.uint8, device=dev)

    # the uploads and zero-fills above ran on torch's current stream; the library's own stream (non-blocking) and RCCL's consume
    # them without any ordering in between: settle them before the first step
    torch.cuda.synchronize()
    first = [True]
    # N > 1: the all-gather is enqueued on the context's own stream (stream-ordered with the kernels round it, no host sync)
    ext = torch.cuda.ExternalStream(ctx.stream_ptr(), device=dev) if multi else None

    done = [0]                                          # steps run so far (the clip position of the resident reference)
    sweeps = []                                         # --exact: sweeps of each slice search

    def step(k):
        done[0] = k + 1
        Y, U, V = src[1 + (k % (nframes - 1))]
        ctx.cur_bind(Y.data_ptr(), U.data_ptr(), V.data_ptr())          # the source frame is resident: no copy
        # luma: the 16 quarter-pel planes (search + sub-pel refinement + MC read them all over). Chroma: MC is their only reader and takes
        # 2 x 64 samples per macroblock, so the frame stage computes those eighth-pel samples itself (jmhip_residual_frame) instead of
        # materialising 64 planes per component -- --chroma-planes builds them as JM's ChromaMCBuffer = 1 does (identical results)
        if not multi:
            ctx.interp_luma(0)
            if args.chroma_planes:
                ctx.interp_chroma(0)
        elif n:
            # a rank only needs the sub-pel planes its band can reach: own rows +- (range + predictor reach 6 + slack); the exact form centres its
            # windows on JM's predictors, themselves clamped to the range: twice the range
            reach = 2 * R + 24 if args.exact else R + 16
            ctx.interp_rows(0, row0 * 16 - reach, row1 * 16 + reach, chroma=args.chroma_planes)
        if n and args.exact:
            ctx.p_slice_search(sp, download=False)
            sweeps.append(ctx.slice_passes())
            ctx.slice_to_frame_band([0], first_mb, count_mb)
            ctx.residual_frame(quants)
        elif n:
            if first[0]:
                ctx.me_frame_async(prm, mbs)
                first[0] = False
            else:
                ctx.me_frame_async(prm, None, n)
            ctx.residual_frame(quants)
            if args.deblock:
                if not multi and args.deblock_slices > 1:
                    ctx.deblock_recon(QP, (QP, QP), disable_idc=2, slice_rows=(MBH + args.deblock_slices - 1) // args.deblock_slices)
                elif not multi:
                    ctx.deblock_recon(QP, (QP, QP))
                else:
                    ctx.deblock_recon(QP, (QP, QP), disable_idc=2, slice_rows=band, mb_row0=row0, mb_rows=row1 - row0)
        if not multi:
            ctx.recon_to_ref(0)
        else:
            if n:                                           # a rank without rows (world > picture rows / band) only receives
                ctx.recon_pack_band(sbuf.data_ptr(), rank, band)
            if solo:
                with torch.cuda.stream(ext):
                    gbuf[rank * chunk:(rank + 1) * chunk].copy_(sbuf)
            elif rehearsal:
                ctx.sync()
                gc = torch.empty(gbuf.shape, dtype=gbuf.dtype)
                dist.all_gather_into_tensor(gc, sbuf.cpu())
                gbuf.copy_(gc)
                torch.cuda.synchronize()
            else:
                with torch.cuda.stream(ext):
                    dist.all_gather_into_tensor(gbuf, sbuf)
            ctx.ref_unpack_bands(0, gbuf.data_ptr(), world, band)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    # timed region: HIP events only round the dominant kernel (roofline.avg_launch_ms); the per-stage breakdown comes from
    # three extra, untimed steps afterwards, so that eleven more event records per step do not sit in the measurement
    ctx.timing_enable(True)
    ctx.timing_select(["me_int"])
    ctx.timing_read()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    fence()
    elapsed = time.perf_counter() - t0
    stage_me = ctx.timing_read()
    ctx.timing_select(list(pkg.STAGES))
    prev_ref_host = None
    for k in range(3):
        if k == 2 and world == 1 and args.cpu_mbs > 0:          # the reference picture the last step searches: kept for the parity check
            ctx.sync()
            ry, _, _, _, _ = ctx.ref_device_planes_ro(0)
            prev_ref_host = np.zeros((H, W), np.uint8)
            ctx.copy_from_device(ry, prev_ref_host)
        step(args.warmup + args.steps + k)
    fence()
    stage = ctx.timing_read()
    stage["me_int"] = stage_me["me_int"]
    ctx.timing_enable(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # checksum of the reference picture the last step produced: the same for every N (strong scaling: the same frames are coded,
    # only sharded differently) -- tests/test_bench_ranks.py compares N = 1 with a rehearsed N = 2
    ctx.sync()
    ry, ru, rv, _, _ = ctx.ref_device_planes_ro(0)
    ref_sum = 0
    for ptr, rows, cols in ((ry, H, W), (ru, H // 2, W // 2), (rv, H // 2, W // 2)):
        host = np.zeros((rows, cols), np.uint8)
        ctx.copy_from_device(ptr, host)
        ref_sum = (ref_sum * 1000003 + int(host.astype(np.int64).sum()) + int((host.astype(np.int64) * (np.arange(cols) % 251 + 1)).sum())) % (1 << 61)

    if rank == 0 or solo:
        total_mbs = MBW * MBH * args.steps
        ms_step = elapsed / args.steps * 1e3
        me_ms, me_launches = stage["me_int"]
        me_avg_ms = me_ms / max(1, me_launches)
        achieved = ME_BYTES_PER_MB * n / (me_avg_ms * 1e-3) / 1e9 if me_launches else 0.0
        sad_ops = (2 * R + 1) ** 2 * 256 * n / (me_avg_ms * 1e-3) if me_launches else 0.0
        valu_floor_ms = n * ((2 * R + 1) ** 2 / 64.0) * VALU_NS_PER_WAVE_CANDIDATE / N_SIMD * 1e-6
        workload = ("1920x1080 (coded 1920x1088, 8160 MBs)" if args.size == "1080p" else "3840x2160 (32400 MBs)") + \
            " YUV420 P-frames, baseline tools, FullSearch +-32, 41 partitions, SAD full-pel + SATD sub-pel, dct_4x4 + dct_chroma, QP %d, 1 reference; " % QP + \
            "16 luma quarter-pel planes per reference, chroma eighth-pel samples " + ("from 2 x 64 planes; " if args.chroma_planes else "computed in MC; ") + \
            ("predictor field (16,-16)+U{-8..8} qpel per MB" if args.pred == "one" else "JM's own predictor per PARTITION (jmhip_p_slice_search of the first picture)") + \
            ("" if args.clip == "translation" else "; ADVERSARIAL i.i.d. noise clip") + \
            ("; + in-loop deblocking (NOT the metric's path)" if args.deblock else "")
        out = {
            "metric": "macroblocks/sec (full-search ME + DCT/quant), %s; bit-exact MV+coeff vs JM" % (
                "1080p" if args.size == "1080p" else "2160p (BASELINE configs[3]: the 4K picture sharded by slices; the metric's own 1080p is the --gpus 1 default)"),
            "metric_note": "predictors are an INPUT of this pipeline (one per macroblock, SURVEY 8(d)'s independent-MBs recipe); `parity_check` compares a sample of the "
                           "last step with the oracle; `slice_search` gives the JM-exact form (predictors + decision on the device, raster-order dependencies kept)",
            "value": round(total_mbs / elapsed, 1), "unit": "macroblocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload,
                       "slices": world, "parallelism": "slice%d" % world},
            "roofline": {"kernel": "me_int_pair_kernel (integer full search, all 41 partitions)", "bound": "valu",
                         "achieved": round(sad_ops / 1e12, 3), "peak": round(sad_ops / 1e12 * me_avg_ms / valu_floor_ms, 3) if me_launches else None,
                         "unit": "T abs-diff/s", "frac": round(valu_floor_ms / me_avg_ms, 4) if me_launches else None,
                         "traffic": pmc_traffic("me_int_pair_kernel", n), "avg_launch_ms": round(me_avg_ms, 4), "units_per_launch": n,
                         "peak_is": "a MODEL, not a guide constant: the time the search's irreducible work -- per candidate 64 v_sad_u8, 25 + 40 adds, "
                                    "20.5 v_min3 -- takes at issue rates measured on this chip (tools/ubench/valu_rate2.hip -> "
                                    "profiles/r01_valu_issue_rates.txt: v_sad_u8 / v_min3_u32 1.89 ns, v_add_u32 1.04 ns per wave-instruction per SIMD, "
                                    "1024 SIMDs); full search is about 1e3 integer ops per compulsory byte (SURVEY 8(d)), so HBM is not the roof that binds",
                         "valu_model": {"model_floor_ms": round(valu_floor_ms, 4), "ns_per_wave_candidate_row": round(VALU_NS_PER_WAVE_CANDIDATE, 1), "simds": N_SIMD},
                         "hbm": {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                                 "algorithmic_bytes_per_unit": ME_BYTES_PER_MB,
                                 "traffic_note": "traffic = (FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch from profiles/r02_final_pmc_*.csv, raw: the guide's x2 "
                                                 "FETCH_SIZE correction is calibrated for 16 B/lane streaming reads and this kernel reads dwords, so it is not applied"}},
            "stages_ms_per_launch": {k: round(v[0] / max(1, v[1]), 4) for k, v in stage.items()},
            "ref_checksum": ref_sum,
        }
        if world == 1:
            # the HBM-bound stages against the same 8 TB/s, from SURVEY 8(d)'s algorithmic bytes at this build's 8-bit samples:
            # luma planes (1 read + 16 written) x padded plane; chroma 2 x (1 + 64) x padded plane; transform/quant/recon per macroblock
            # cur + pred 768 B in, levels 384 x 4 B + recon 384 B out
            ms = out["stages_ms_per_launch"]
            alg = {"interp_luma": (W + 40) * (H + 40) * 17, "interp_chroma": 2 * (W // 2 + 20) * (H // 2 + 20) * 65,
                   "mc+tq": n * (768 + 384 * 4 + 384)}
            t_ms = {"interp_luma": ms["interp_luma"], "interp_chroma": ms["interp_chroma"], "mc+tq": ms["mc"] + ms["tq"]}
            out["hbm_bound_stages"] = {k: {"algorithmic_bytes": alg[k], "ms": round(t_ms[k], 4), "achieved_GBs": round(alg[k] / (t_ms[k] * 1e-3) / 1e9, 1),
                                           "frac_of_8TBs": round(alg[k] / (t_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)} for k in alg if t_ms[k] > 0}
        if world == 1 and args.cpu_mbs > 0:
            ref = cpu_baseline_reference(frames)
            port = cpu_baseline(pkg, frames, args.cpu_mbs)
            out["cpu_baseline"] = ref if ref is not None else port
            out["cpu_baseline_port"] = port
            out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if world == 1 and args.cpu_mbs > 0:
            out["parity_check"] = parity_check(pkg, ctx, frames, mbs, prm, last_src=1 + ((args.warmup + args.steps + 2) % (nframes - 1)), prev_ref=prev_ref_host)
            if not args.deblock:
                # for information: the same step with JM's default in-loop deblocking filter in the loop (jmhip_deblock_recon, relaxation schedule)
                args.deblock = True
                k0 = done[0]
                for k in range(3):
                    step(k0 + k)
                fence()
                t0 = time.perf_counter()
                for k in range(10):
                    step(k0 + 3 + k)
                fence()
                dt = (time.perf_counter() - t0) / 10
                args.deblock = False
                out["with_loop_filter"] = {"ms_per_step": round(dt * 1e3, 4), "macroblocks_per_s": round(MBW * MBH / dt, 1),
                                           "note": "LoopFilterDisable = 0: every reference picture is deblocked on the device before it is interpolated; not the metric's path"}
            # the clip's pictures that follow the reference the last step left, in display order
            order = [1 + ((done[0] + j) % (nframes - 1)) for j in range(nframes - 1)]
            out["slice_search"] = slice_search_times(pkg, ctx, lam, src, order)
            out["per_partition_pred"] = per_partition_predictors(pkg, ctx, lam, prm, mbs)
            e2e = jm_end_to_end(frames)
            if e2e is not None:
                out["jm_end_to_end"] = e2e
            e2e = jm_end_to_end(frames, config3=True)
            if e2e is not None:
                out["jm_end_to_end_config3_tools"] = e2e
            e2e = jm_end_to_end(frames, rdopt1=True)
            if e2e is not None:
                out["jm_end_to_end_rdopt1_speculative"] = e2e
            # for information: the same step in its JM-EXACT form (`--exact`: the slice search with JM's own predictors and decision instead of one
            # synthetic predictor per macroblock, then the frame stage from that decision), in a process of its own
            try:
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--exact", "--steps", "8", "--warmup", "3", "--cpu-mbs", "0"], capture_output=True, text=True, timeout=300)
                ex = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["jm_exact_pipeline"] = {"macroblocks_per_s": ex["value"], "ms_per_step": ex["ms_per_step"], "slice_search_ms": ex["roofline"]["slice_search_ms"],
                                            "sweeps_last_steps": ex["roofline"]["sweeps_last_steps"], "ref_checksum": ex.get("ref_checksum"),
                                            "note": "bench.py --exact: jmhip_p_slice_search (FullSearch +-32 round every partition's own predictor, sub-pel, skip shortcut, low-complexity "
                                                    "decision, relaxation sweeps) -> jmhip_slice_to_frame_band -> jmhip_residual_frame -> next reference; what the metric's workload "
                                                    "costs when JM's raster-order dependencies are kept"}
            except Exception as e:
                sys.stderr.write("jm_exact_pipeline: %s\n" % e)
        if args.exact:
            out["metric_note"] = ("--exact: the JM-EXACT form of the step -- jmhip_p_slice_search per slice (predictors, FullSearch +-32 per partition round its own "
                                  "predictor, sub-pel, skip shortcut, low-complexity decision; relaxation sweeps to JM's fixpoint), jmhip_slice_to_frame_band, "
                                  "jmhip_residual_frame, band exchange. The metric's own workload is the default run (independent macroblocks)")
            out["config"]["workload"] = workload.replace("predictor field (16,-16)+U{-8..8} qpel per MB", "JM's own predictors (SetMotionVectorPredictor in raster order, on the device)")
            out["config"]["slices"] = max(world, args.exact_slices or 1)
            me_ms, me_launches = stage["me_int"]
            out["roofline"] = {"kernel": "jmhip_p_slice_search (x_sim_kernel + me_int_pair_kernel<list> + me_sub_kernel<list>, sweeps to the fixpoint)", "bound": "valu",
                               "achieved": round((2 * R + 1) ** 2 * 256 * n / (me_ms / max(1, me_launches) * 1e-3) / 1e12, 3) if me_launches else None,
                               "peak": round((2 * R + 1) ** 2 * 256 * n / (valu_floor_ms * 1e-3) / 1e12, 3), "unit": "T abs-diff/s",
                               "frac": round(valu_floor_ms / (me_ms / max(1, me_launches)), 4) if me_launches else None, "traffic": None,
                               "note": "achieved = ONE exhaustive search per macroblock's worth of abs-diffs over the whole slice search's time (its sweeps search "
                                       "several times that: every record whose predictor changed); peak = the same VALU model as the default line",
                               "slice_search_ms": round(me_ms / max(1, me_launches), 3), "sweeps_last_steps": sweeps[-4:]}
        if solo:
            out = {"DIAGNOSTIC_solo_rank": solo, "ms_per_step_of_this_rank": out["ms_per_step"], "stages_ms_per_launch": out.get("stages_ms_per_launch")}
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
