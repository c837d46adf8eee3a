"""The drop-in claim, end to end: the REAL JM encoder with its hot path bound to libjmhip.so (integration/jm_shim.c,
built as oracle/_ref/jm_hip) must write the same bitstream and reconstruction, byte for byte, as the unmodified
encoder (oracle/_ref/jm_plain) on the same clip and configuration.

Both binaries are compiled in the build container from the reference sources where they lie and travel to the GPU
box under oracle/_ref/ (git-ignored). The clip is synthetic (the reference's own clips do not travel); the cfg is
written here, every key a JM configuration key. JMHIP_SHIM_STATS shows how many calls the device served.
"""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RDIR = os.path.join(ROOT, "oracle", "_ref")
HAVE = os.path.exists(os.path.join(RDIR, "jm_hip")) and os.path.exists(os.path.join(RDIR, "jm_plain"))

CFG = """
InputFile = "clip.yuv"
InputHeaderLength = 0
StartFrame = 0
FramesToBeEncoded = {frames}
FrameRate = 30.0
SourceWidth = {w}
SourceHeight = {h}
OutputFile = "out.264"
ReconFile = "out_rec.yuv"
TraceFile = "trace_enc.txt"
ProfileIDC = {profile}
LevelIDC = 40
IntraPeriod = 0
QPISlice = {qp}
QPPSlice = {qp}
QPBSlice = {qp}
SearchRange = {R}
NumberReferenceFrames = {refs}
NumberBFrames = {bframes}
FrameSkip = {bframes}
BiPredMotionEstimation = {bipred}
BiPredMERefinements = 3
BiPredMESearchRange = 16
BiPredMESubPel = 2
WeightedBiprediction = {wbp}
WeightedPrediction = {wp}
UseWeightedReferenceME = {wp}
SymbolMode = {cabac}
SearchMode = {search}
RDOptimization = {rdopt}
MEDistortionFPel = {fpel}
MEDistortionHPel = {hpel}
MEDistortionQPel = {qpel}
MDDistortion = {mdm}
ChromaMCBuffer = 1
ChromaMEEnable = {cme}
ChromaMEWeight = {cmw}
RestrictSearchRange = 2
AdaptiveRounding = {adrnd}
Transform8x8Mode = {t8x8}
YUVFormat = {yuv}
LoopFilterParametersFlag = {lfflag}
LoopFilterDisable = {lfidc}
LoopFilterAlphaC0Offset = {lfa}
LoopFilterBetaOffset = {lfb}
SliceMode = {slicemode}
SliceArgument = {slicearg}
DisableIntraInInter = {noi}
{extra}
"""

# Scaling matrices (a18): a q_matrix file in JM's format (src/q_matrix.c:31-49 names the lists) with this test's own values -- the reference's
# bin/q_matrix.cfg does not travel --, present in the SPS for the six 4x4 lists and the two 8x8 luma lists. JM then builds LevelScale / InvLevelScale
# as (quant_coef << 4) / M and dequant_coef * M (src/q_matrix.c:451-738), which the binding hands to the device per call.
QM_KEYS = "QmatrixFile = \"qm.cfg\"\nScalingMatrixPresentFlag = 1\n" + "".join("ScalingListPresentFlag%d = 1\n" % i for i in range(8))


def write_qmatrix(path):
    names4 = ["INTRA4X4_LUMA", "INTRA4X4_CHROMAU", "INTRA4X4_CHROMAV", "INTER4X4_LUMA", "INTER4X4_CHROMAU", "INTER4X4_CHROMAV"]
    with open(path, "w") as f:
        for k, name in enumerate(names4):
            m = [[min(255, 8 + k + (3 + (k & 1)) * (i + j) + (2 if i == j else 0)) for i in range(4)] for j in range(4)]
            f.write("%s =\n%s\n\n" % (name, ",\n".join(",".join("%d" % v for v in row) for row in m)))
        for k, name in enumerate(["INTRA8X8_LUMA", "INTER8X8_LUMA"]):
            m = [[min(255, 9 + 3 * k + 2 * (i + j) + (i * j) // 3) for i in range(8)] for j in range(8)]
            f.write("%s =\n%s\n\n" % (name, ",\n".join(",".join("%d" % v for v in row) for row in m)))

CASES = {
    # name: cfg values                                                                      what it exercises
    "full_baseline": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=1, rdopt=1, adrnd=1, yuv=1),  # FullPel+SubPel, dct_4x4/16x16/chroma
    "fastfull_high": dict(search=0, profile=100, cabac=1, t8x8=1, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1),  # FastFull, dct_8x8, B slices, 2 refs
    "fastfull_lowcplx": dict(search=0, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1),  # FastFull pos_00 pre-check, search_range/2 on ref 1
    # B slices with bi-predictive ME (16x16): FullPelBlockMotionBiPred x4 refinements + SubPelBlockSearchBiPred x2, plain and weighted
    "main_bipred": dict(search=-1, profile=77, cabac=1, t8x8=0, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1, bipred=1),
    "high_bipred_weighted_t8": dict(search=0, profile=100, cabac=1, t8x8=1, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1, bipred=1, wbp=2),
    # EPZS / UMHexagonS: JM's own walker, integer-pel computeSAD / computeSATD answered from the device's distortion surfaces
    "epzs_main": dict(search=3, profile=77, cabac=1, t8x8=0, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1),
    "umhex_baseline": dict(search=1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=1, adrnd=1, yuv=1),
    "epzs_satd_high": dict(search=3, profile=100, cabac=1, t8x8=1, bframes=0, refs=1, rdopt=1, adrnd=0, yuv=1, fpel=2),
    # BASELINE config 5 in small: 4:2:2, UMHexagonS, explicit weighted prediction used in ME (computeSADWP surfaces), fading clip
    "umhex_wp_422": dict(search=1, profile=122, cabac=1, t8x8=1, bframes=0, refs=2, rdopt=1, adrnd=0, yuv=2, wp=1, fade=1),
    # explicit weighted prediction used in ME on a fading clip with the exhaustive searches: the device weights the reference window /
    # the sub-pel rows (computeSADWP / computeSATDWP); P only, and B slices with explicit bi-prediction weights
    "full_wp": dict(search=-1, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=1, adrnd=0, yuv=1, wp=1, fade=1),
    "fastfull_wp_b": dict(search=0, profile=100, cabac=1, t8x8=1, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1, wp=1, wbp=1, fade=1),
    # RD-off decision with the 8x8 transform: TransformDecision and GetSkipCostMB costs from the device
    "lowcplx_t8_decision": dict(search=0, profile=100, cabac=0, t8x8=1, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1),
    # in-loop deblocking: three slices per picture, filter kept inside slices (idc 2), non-zero alpha / beta offsets, coarse quantiser
    "slices_deblock_idc2": dict(search=0, profile=100, cabac=1, t8x8=1, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1, qp=38, lfflag=1, lfidc=2, lfa=2, lfb=-1, slicemode=1),
    # slices that begin in the middle of a macroblock row (27 macroblocks, 11 per row): left / top availability differs inside a row
    "slices_midrow_idc2": dict(search=0, profile=77, cabac=1, t8x8=0, bframes=0, refs=1, rdopt=1, adrnd=0, yuv=1, qp=36, lfflag=1, lfidc=2, lfa=0, lfb=0, slicemode=1, slicearg=27),
    "slices_deblock_across_422": dict(search=0, profile=122, cabac=0, t8x8=1, bframes=0, refs=1, rdopt=1, adrnd=0, yuv=2, qp=36, lfflag=1, lfidc=0, lfa=-2, lfb=3, slicemode=1),
    # 4:4:4 (High 4:4:4 Predictive): chroma planes take the luma filter in the loop filter, quarter-pel chroma planes, dct_4x4 on all three planes
    "fastfull_444": dict(search=0, profile=244, cabac=1, t8x8=1, bframes=0, refs=1, rdopt=1, adrnd=0, yuv=3, qp=34),
    # every error metric of computeUniPred and the chroma term through the general search (me_metric.hip): SSE everywhere with Cb / Cr at
    # integer and sub-pel positions; Hadamard SAD at integer positions too (8x8 for block types 1..4; the refinements start from the carried
    # minimum); FastFullSearch with the unweighted chroma term of SetupFastFullPelSearch and SAD half-pel; weighted reference ME with chroma
    "full_sse_chroma2": dict(search=-1, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=1, adrnd=0, yuv=1, fpel=1, hpel=1, qpel=1, cme=2, cmw=2),
    "full_satd_all_t8": dict(search=-1, profile=100, cabac=1, t8x8=1, bframes=1, refs=2, rdopt=0, adrnd=0, yuv=1, fpel=2, hpel=2, qpel=2),
    "fastfull_chroma1": dict(search=0, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, fpel=0, hpel=0, qpel=2, cme=1, cmw=1),
    "fastfull_satd_fpel_422": dict(search=0, profile=122, cabac=1, t8x8=1, bframes=0, refs=1, rdopt=1, adrnd=0, yuv=2, fpel=2, hpel=2, qpel=2, cme=2, cmw=1),
    "full_wp_chroma": dict(search=-1, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=1, adrnd=0, yuv=1, wp=1, fade=1, fpel=0, hpel=0, qpel=2, cme=2, cmw=1),
    "full_wp_sse_chroma": dict(search=-1, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=1, adrnd=0, yuv=1, wp=1, fade=1, fpel=1, hpel=1, qpel=1, cme=2, cmw=1),
    "full_lowcplx_422": dict(search=-1, profile=122, cabac=0, t8x8=1, bframes=0, refs=1, rdopt=0, adrnd=1, yuv=2),  # rdopt off centre rule, 4:2:2 chroma DC
    # scaling matrices (q_matrix.c:451-738): non-flat LevelScale / InvLevelScale tables for every 4x4 list and the 8x8 luma lists, I / P / B, adaptive rounding on
    "fastfull_high_qmatrix": dict(search=0, profile=100, cabac=1, t8x8=1, bframes=1, refs=2, rdopt=1, adrnd=1, yuv=1, qmatrix=1),
    "full_high_qmatrix_cavlc_lowcplx": dict(search=-1, profile=100, cabac=0, t8x8=1, bframes=0, refs=1, rdopt=0, adrnd=0, yuv=1, qmatrix=1, qp=34),
}


def make_clip(path, w, h, frames, yuv, fade=0, diverse=0):
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:h + 64, 0:w + 64]
    base = ((np.sin(xx / 7.0) * np.cos(yy / 5.0)) * 60 + 128 + rng.normal(0, 10, (h + 64, w + 64))).clip(0, 255)
    cw, ch = (w // 2, h // 2) if yuv == 1 else (w // 2, h) if yuv == 2 else (w, h)
    with open(path, "wb") as f:
        for t in range(frames):
            dx, dy = 3 * t, 2 * t
            gain = 1.0 - 0.07 * t * fade              # a fade gives the explicit weights something to estimate
            luma = base[16 + dy:16 + dy + h, 16 + dx:16 + dx + w] + rng.normal(0, 2, (h, w))
            if diverse:                               # regions with their own motion: a static third (zero vectors next to moving neighbours: the skip
                x1, x2 = (w // 3) & ~15, (2 * w // 3) & ~15      # vector's rule, mv-search.c:1189) and one moving the other way
                luma[:, :x1] = base[16:16 + h, 16:16 + x1] + rng.normal(0, 1, (h, x1))
                luma[h // 2:, x2:] = base[16 - t + h // 2:16 - t + h, 16 - 2 * t + x2:16 - 2 * t + w] + rng.normal(0, 2, (h - h // 2, w - x2))
            f.write((luma * gain).clip(0, 255).astype(np.uint8).tobytes())
            for k in range(2):
                c = base[8 + dy // 2:8 + dy // 2 + ch, 8 + dx // 2 + 5 * k:8 + dx // 2 + 5 * k + cw]
                f.write((c * 0.5 + 64).clip(0, 255).astype(np.uint8).tobytes())


def run(exe, d, env=None):
    for f in ("out.264", "out_rec.yuv"):
        if os.path.exists(os.path.join(d, f)):
            os.remove(os.path.join(d, f))
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([os.path.join(RDIR, exe), "-d", "case.cfg"], cwd=d, env=e, capture_output=True, text=True, timeout=900)
    # JM's main() returns a nonzero status even on success: judge by the files it wrote
    assert os.path.exists(os.path.join(d, "out.264")), (r.stdout[-800:], r.stderr[-800:])
    with open(os.path.join(d, "out.264"), "rb") as a, open(os.path.join(d, "out_rec.yuv"), "rb") as b:
        return a.read(), b.read(), r.stderr


def prepare(tmp_path, name, w=176, h=144, frames=3, R=16, qp=28):
    v = dict(dict(w=w, h=h, frames=frames, R=R, qp=qp, lfflag=0, lfidc=0, lfa=0, lfb=0, slicemode=0, slicearg=33), **CASES[name])
    v.setdefault("fpel", 0)
    v.setdefault("hpel", 2)
    v.setdefault("qpel", 2)
    v.setdefault("cme", 0)
    v.setdefault("mdm", 2)
    v.setdefault("cmw", 1)
    v.setdefault("bipred", 0)
    v.setdefault("wbp", 0)
    v.setdefault("wp", 0)
    v.setdefault("noi", 0)
    v["extra"] = QM_KEYS if v.get("qmatrix") else ""
    if v.get("qmatrix"):
        write_qmatrix(tmp_path / "qm.cfg")
    with open(tmp_path / "case.cfg", "w") as f:
        f.write(CFG.format(**v))
    make_clip(tmp_path / "clip.yuv", w, h, frames + v["bframes"] * (frames - 1), v["yuv"], v.get("fade", 0), v.get("diverse", 0))


@pytest.mark.reference
@pytest.mark.skipif(not HAVE, reason="oracle/_ref/jm_hip not built (container: make -C oracle ref)")
def test_shim_forwards_everything_when_masked_off(tmp_path):
    """Host-side check, no GPU: with every group masked off the shim must be transparent (link order, forwarding)."""
    prepare(tmp_path, "full_baseline", frames=2)
    want = run("jm_plain", tmp_path)
    got = run("jm_hip", tmp_path, {"JMHIP_SHIM": "0", "JMHIP_SHIM_STATS": "1"})
    assert got[0] == want[0] and got[1] == want[1]
    assert len(want[0]) > 500
    assert re.search(r"FullPelBlockMotionSearch\s+device\s+0\s+forwarded\s+[1-9]", got[2]), got[2]


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="oracle/_ref/jm_hip did not travel")
@pytest.mark.parametrize("name", [n for n in CASES if not n.startswith("slice_")])
def test_jm_with_hip_hot_path_is_byte_identical(tmp_path, name):
    prepare(tmp_path, name)
    want = run("jm_plain", tmp_path)
    got = run("jm_hip", tmp_path, {"JMHIP_SHIM_STATS": "1"})
    stats = got[2]
    assert got[0] == want[0], "bitstream differs\n" + stats
    assert got[1] == want[1], "reconstruction differs\n" + stats
    served = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"(\w+)\s+device\s+(\d+)\s+forwarded\s+(\d+)", stats)}
    print(name, served)
    # the device must actually have served the path (not a forward-everything pass)
    if CASES[name]["yuv"] == 3:
        # 4:4:4: JM interpolates and searches all three planes through its luma functions with per-plane globals; the binding leaves
        # the search in JM there and serves the luma planes, the transforms of the three planes and the loop filter
        assert served["getSubImagesLuma"][0] > 0 and served["dct_4x4"][0] > 1000 and served["DeblockFrame"][0] >= 2 and served["DeblockFrame"][1] == 0
        return
    assert served["getSubImagesLuma"][0] > 0 and served["getSubImagesChroma"][0] > 0
    if CASES[name]["search"] in (-1, 0):
        assert served["SubPelBlockMotionSearch"][0] > 1000 and served["SubPelBlockMotionSearch"][1] == 0
        key = "FullPelBlockMotionSearch" if CASES[name]["search"] == -1 else "FastFullPelBlockMotionSearch"
        assert served[key][0] > 1000 and served[key][1] == 0
    else:
        assert served["EPZS_UMHex_integer_walks"][0] > 1000
        kernel = "computeSATD" if CASES[name].get("fpel") == 2 else "computeSAD"
        assert served[kernel][0] > 10000, "integer-pel distortions were not served from the device surface"
    assert served["dct_4x4"][0] > 1000 and served["dct_chroma"][0] > 100
    assert served["DeblockFrame"][0] >= 2 and served["DeblockFrame"][1] == 0, "the in-loop filter did not run on the device"
    if CASES[name]["t8x8"]:
        assert served["dct_8x8"][0] > 100
    if name == "lowcplx_t8_decision":
        assert served["TransformDecision"][0] > 100 and served["GetSkipCostMB"][0] > 50
    if CASES[name]["bframes"] and name != "epzs_main":
        assert served["BIDPartitionCost"][0] > 500, "bi-directional partition costs were not served by the device"
    if CASES[name].get("bipred"):
        assert served["FullPelBlockMotionBiPred"][0] > 100 and served["FullPelBlockMotionBiPred"][1] == 0
        assert served["SubPelBlockSearchBiPred"][0] > 50 and served["SubPelBlockSearchBiPred"][1] == 0


# ------------------------------------------------------------------ slice-level binding (jm_shim.c, mask 0x1000)
# Low-complexity mode with intra off in P slices: the whole motion search + inter decision of a P slice is ONE jmhip_p_slice_search call and
# every BlockMotionSearch call is answered from its records (after a predictor check). The EPZS / UMHexagonS walkers, their sub-pel searches
# and the predictor run on the device: JM's own are never called.
SLICE_CASES = {
    "slice_full": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=1, rdopt=0, adrnd=1, yuv=1, noi=1),
    "slice_fastfull_2ref": dict(search=0, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1),
    "slice_epzs_2ref": dict(search=3, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1),            # BASELINE config 3's search in small
    "slice_epzs_satd_fpel": dict(search=3, profile=77, cabac=1, t8x8=0, bframes=0, refs=1, rdopt=0, adrnd=0, yuv=1, noi=1, fpel=2),
    "slice_umhex_2ref": dict(search=1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, noi=1),
    "slice_umhex_wp_422": dict(search=1, profile=122, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=2, wp=1, fade=1, noi=1),   # config 5 in small
    # Transform8x8Mode in the slice binding (BASELINE config 3 says "8x8 transform enabled"): both transform sizes compete (1) -- the device also
    # quantises the 8x8-transform P8x8 pass, whose coded-block pattern decides the partitioning -- and 8x8 only (2)
    "slice_epzs_t8_high": dict(search=3, profile=100, cabac=1, t8x8=1, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, qp=36),
    "slice_umhex_t8_cavlc": dict(search=1, profile=100, cabac=0, t8x8=1, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, qp=32),
    "slice_fastfull_t8_q40": dict(search=0, profile=100, cabac=0, t8x8=1, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, qp=40),
    "slice_full_t8only": dict(search=-1, profile=100, cabac=1, t8x8=2, bframes=0, refs=1, rdopt=0, adrnd=0, yuv=1, noi=1),
    "slice_umhexsmp_2ref": dict(search=2, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, noi=1),               # the simplified UMHexagonS
    "slice_umhexsmp_t8_satd": dict(search=2, profile=100, cabac=1, t8x8=1, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, qp=34, fpel=2),
    # fixed-size slices with a stateless search: ALL slices of a picture in one device call (slice_mbs); 27 macroblocks per slice, 11 per row
    "slice_full_four_slices_one_call": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, noi=1, slicemode=1, slicearg=27, lfflag=1, lfidc=2),
    "slice_umhexsmp_slices_one_call_t8": dict(search=2, profile=100, cabac=1, t8x8=1, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, slicemode=1, slicearg=40, qp=34),
    "slice_epzs_four_slices_midrow": dict(search=3, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, slicemode=1, slicearg=27, lfflag=1, lfidc=2),
    # NumberReferenceFrames = 5, as in every cfg the reference ships (bin/encoder_baseline.cfg:53): seven pictures, so that list 0 grows to five entries
    "slice_fastfull_5ref": dict(search=0, profile=66, cabac=0, t8x8=0, bframes=0, refs=5, rdopt=0, adrnd=1, yuv=1, noi=1, frames=7),
    "slice_epzs_5ref": dict(search=3, profile=77, cabac=1, t8x8=0, bframes=0, refs=5, rdopt=0, adrnd=0, yuv=1, noi=1, frames=7),
    # SSE (MEDistortion* / ModeDecisionMetric 1) in the slice path: the walkers at every level, the exhaustive searches at sub-pel positions, the decision costs
    "slice_epzs_sse": dict(search=3, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=1, fpel=1, hpel=1, qpel=1, mdm=1),
    "slice_umhex_sse_hpel": dict(search=1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, noi=1, fpel=0, hpel=1, qpel=2),
    "slice_full_sse_subpel_t8": dict(search=-1, profile=100, cabac=1, t8x8=1, bframes=0, refs=1, rdopt=0, adrnd=0, yuv=1, noi=1, fpel=0, hpel=1, qpel=1, mdm=1, qp=32),
    "slice_full_range40": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, noi=1, R=40),      # the widest range of the sweeps over the frame kernels
    "slice_umhex_5ref_t8": dict(search=1, profile=100, cabac=1, t8x8=1, bframes=0, refs=5, rdopt=0, adrnd=0, yuv=1, noi=1, frames=7, qp=32),
}
CASES.update(SLICE_CASES)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="oracle/_ref/jm_hip did not travel")
@pytest.mark.parametrize("name", list(SLICE_CASES))
def test_jm_slice_level_binding_is_byte_identical(tmp_path, name):
    nframes = SLICE_CASES[name].get("frames", 4)
    prepare(tmp_path, name, frames=nframes, R=SLICE_CASES[name].get("R", 16))
    want = run("jm_plain", tmp_path)
    got = run("jm_hip", tmp_path, {"JMHIP_SHIM_STATS": "1"})
    stats = got[2]
    assert got[0] == want[0], "bitstream differs\n" + stats
    assert got[1] == want[1], "reconstruction differs\n" + stats
    m = re.search(r"^\s*BlockMotionSearch\s+device\s+(\d+)\s+forwarded\s+(\d+)", stats, re.M)
    sl = re.search(r"P slices \(one device call each\)\s+device\s+(\d+)", stats)
    info = re.search(r"slice binding: (\d+) slices, (\d+) kernel passes", stats)
    print(name, m.groups(), sl.groups(), info.groups())
    nslices = 1          # also with four slices of 27 macroblocks per picture: all slices of a picture go in ONE device call (slice_mbs)
    assert int(sl.group(1)) == (nframes - 1) * nslices, "one device call per P slice, or per picture where the slices go in one call"
    per_mb = {0: 41, 1: 45, 2: 9}[SLICE_CASES[name]["t8x8"]]      # Transform8x8Mode 1: four more calls (the 8x8-transform P8x8 pass); 2: modes 1..3 + that pass only
    assert int(m.group(1)) >= (nframes - 1) * 99 * per_mb and int(m.group(2)) == 0, "every BlockMotionSearch call of the P pictures must be served from the slice records"
    # JM's own search functions must not have run at all in the P pictures
    for sym in ("FullPelBlockMotionSearch", "FastFullPelBlockMotionSearch", "SubPelBlockMotionSearch", "EPZS_UMHex_integer_walks", "computeSAD", "computeSATD"):
        mm = re.search(r"^\s*%s\s+device\s+(\d+)\s+forwarded\s+(\d+)" % sym, stats, re.M)
        assert mm and int(mm.group(1)) == 0 and int(mm.group(2)) == 0, (sym, mm and mm.groups())
    # 4:2:0 with the 4x4 transform: the frame stage is bound at slice level too (mask 0x4000) -- every prediction and every dct_4x4 / dct_chroma of
    # the P pictures is answered from the device's prediction picture and per-macroblock records, none is computed by JM
    served = {k: tuple(int(v) for v in re.search(r"^\s*%s\s+device\s+(\d+)\s+forwarded\s+(\d+)" % re.escape(k), stats, re.M).groups())
              for k in ("frame stage of P slices", "dct_4x4 (slice records)", "dct_chroma (slice records)", "LumaPrediction (slice)", "ChromaPrediction4x4 (slice)")}
    mbs = (nframes - 1) * 99
    if SLICE_CASES[name]["yuv"] == 1 and SLICE_CASES[name]["t8x8"] == 0:
        assert served["frame stage of P slices"][0] == nframes - 1, served
        # -- JM predicts and transforms every macroblock twice: its P8x8 candidate inside submacroblock_mode_decision (src/mode_decision.c:874), then the
        # decided mode; both passes are answered (the candidate from a second frame-stage pass, jmhip_slice_to_frame_candidates)
        # (JM leaves the candidate pass out for some macroblocks, so the luma counts lie between one and two passes; nothing is left to JM)
        assert 16 * mbs <= served["dct_4x4 (slice records)"][0] <= 32 * mbs and served["dct_4x4 (slice records)"][1] == 0 and served["dct_chroma (slice records)"] == (2 * mbs, 0), served
        assert served["LumaPrediction (slice)"] == served["dct_4x4 (slice records)"] and served["ChromaPrediction4x4 (slice)"] == (8 * mbs, 0), served
        lazy = re.search(r"^\s*sub-pel planes fetched on demand\s+device\s+(\d+)", stats, re.M)
        assert int(lazy.group(1)) == 0, "the bound P pictures needed no sub-pel plane on the host\n" + stats
    else:
        assert served["frame stage of P slices"][0] == 0 and served["dct_4x4 (slice records)"][0] == 0, served


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="oracle/_ref/jm_hip did not travel")
def test_jm_1080p_full_search_slice_binding_is_byte_identical_and_faster(tmp_path):
    """BASELINE config 2 through the real encoder: 1920x1080, FullSearch +-32, I + P, low-complexity decision. The P picture's whole motion
    search + inter decision is one device call, and so is its frame stage -- prediction, residual, transform, quantisation, reconstruction -- whose
    results answer JM's LumaPrediction / ChromaPrediction4x4 / dct_4x4 / dct_chroma calls (mask 0xd801: sub-pel planes built on the device and left there,
    slice binding with the frame stage, loop filter; everything else stays JM's)."""
    import time
    CASES["slice_full_1080p"] = dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=1, rdopt=0, adrnd=1, yuv=1, noi=1)
    prepare(tmp_path, "slice_full_1080p", w=1920, h=1080, frames=2, R=32)
    t0 = time.perf_counter()
    want = run("jm_plain", tmp_path)
    t_plain = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = run("jm_hip", tmp_path, {"JMHIP_SHIM_STATS": "1", "JMHIP_SHIM": "d801"})
    t_hip = time.perf_counter() - t0
    stats = got[2]
    assert got[0] == want[0] and got[1] == want[1], "1080p encode differs\n" + stats
    m = re.search(r"^\s*BlockMotionSearch\s+device\s+(\d+)\s+forwarded\s+(\d+)", stats, re.M)
    assert int(m.group(1)) == 8160 * 41 and int(m.group(2)) == 0, m.groups()
    d4 = re.search(r"^\s*dct_4x4 \(slice records\)\s+device\s+(\d+)\s+forwarded\s+(\d+)", stats, re.M)
    dc = re.search(r"^\s*dct_chroma \(slice records\)\s+device\s+(\d+)\s+forwarded\s+(\d+)", stats, re.M)
    assert 8160 * 16 <= int(d4.group(1)) <= 8160 * 32 and int(d4.group(2)) == 0 and (int(dc.group(1)), int(dc.group(2))) == (8160 * 2, 0), (d4.groups(), dc.groups())
    print("1080p I+P, FullSearch +-32: jm_plain %.1f s, jm_hip %.1f s; %s of %s BlockMotionSearch calls answered from the slice record (hit rate 100%%)" % (
        t_plain, t_hip, m.group(1), m.group(1)))
    assert t_hip < t_plain


# ------------------------------------------------------------------ speculative slice binding (mask 0x2000): JM's own decision is rate-distortion optimised
SPEC_CASES = {
    "spec_full_rdopt1_intra": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=1, adrnd=1, yuv=1, noi=0),       # JM's defaults: RD decision, intra candidates
    "spec_fastfull_rdopt1_high_t8": dict(search=0, profile=100, cabac=1, t8x8=1, bframes=0, refs=2, rdopt=1, adrnd=1, yuv=1, noi=0),
    "spec_umhexsmp_rdopt2": dict(search=2, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=2, adrnd=0, yuv=1, noi=0),
    "spec_full_lowcplx_with_intra": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=1, rdopt=0, adrnd=1, yuv=1, noi=0),   # rdopt 0 but intra candidates on
    "spec_fastfull_rdopt1_bframes": dict(search=0, profile=77, cabac=1, t8x8=0, bframes=1, refs=2, rdopt=1, adrnd=0, yuv=1, noi=0),     # P pictures bound, B pictures JM's
    # RDOptimization 0 with intra candidates on a clip whose regions move differently (a static third beside moving ones): the 16x16 record went
    # through the skip shortcut, whose vector depends on the neighbours' references and zero vectors, not on the call's predictor -- the binding
    # answers it only when JM's FindSkipModeMotionVector agrees with the device's (jm_shim.c)
    "spec_full_lowcplx_intra_diverse": dict(search=-1, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=1, yuv=1, noi=0, diverse=1),
    "spec_fastfull_lowcplx_intra_diverse": dict(search=0, profile=66, cabac=0, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=0, diverse=1),
    "spec_umhexsmp_lowcplx_intra_diverse": dict(search=2, profile=77, cabac=1, t8x8=0, bframes=0, refs=2, rdopt=0, adrnd=0, yuv=1, noi=0, diverse=1, qp=34),
}
CASES.update(SPEC_CASES)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="oracle/_ref/jm_hip did not travel")
@pytest.mark.parametrize("name", list(SPEC_CASES))
def test_jm_speculative_slice_binding_is_byte_identical(tmp_path, name):
    """RDOptimization 1 / 2 and intra candidates in P slices -- JM's default kind of configuration: the device searches the slice with its
    low-complexity decision as the guess, a BlockMotionSearch call is answered from the record when JM's predictor equals the recorded one and runs
    in JM otherwise. Identical bitstreams whatever the hit rate; the hit rate is what makes it worth it."""
    prepare(tmp_path, name, frames=4)
    want = run("jm_plain", tmp_path)
    got = run("jm_hip", tmp_path, {"JMHIP_SHIM_STATS": "1", "JMHIP_SHIM": "3801"})
    stats = got[2]
    assert got[0] == want[0], "bitstream differs\n" + stats
    assert got[1] == want[1], "reconstruction differs\n" + stats
    m = re.search(r"^\s*BlockMotionSearch\s+device\s+(\d+)\s+forwarded\s+(\d+)", stats, re.M)
    served, fwd = int(m.group(1)), int(m.group(2))
    print(name, "served", served, "forwarded", fwd, "hit rate %.1f %%" % (100.0 * served / max(1, served + fwd)))
    # (with B pictures the forwarded count includes every call of the B slices, which are JM's own)
    assert served > 2000 and (served > fwd or SPEC_CASES[name]["bframes"]), "the speculation should hit for most calls of P pictures"
