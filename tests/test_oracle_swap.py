"""Pins the oracle: the C restatement (oracle/jmo_*.c) is swapped INTO the real JM encoder built from
/root/reference (oracle/tap/swap_oracle.c, symbol interposition) and the bitstream + reconstruction JM
writes must stay byte-identical to the unmodified encoder's, for every shipped cfg x SearchMode and a set
of option variants. The first case is the reference's own known-answer pair bin/test.264 + bin/test_rec.yuv.

Container only: needs /root/reference (sources + bin/ fixtures) and `make -C oracle ref`.
Skipped on the GPU box, where neither exists.
"""
import hashlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
RDIR = os.path.join(ROOT, "oracle", "_ref")

pytestmark = [
    pytest.mark.reference,
    pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "bin")), reason="/root/reference not present"),
]


@pytest.fixture(scope="module")
def rundir(tmp_path_factory):
    if not (os.path.exists(os.path.join(RDIR, "jm_swap")) and os.path.exists(os.path.join(RDIR, "jm_plain"))):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-j8", "all", "ref"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = tmp_path_factory.mktemp("jmrun")
    for f in os.listdir(os.path.join(REF, "bin")):
        if f.endswith(".cfg") or f.endswith(".yuv"):
            shutil.copy(os.path.join(REF, "bin", f), d / f)
    for f in ("test.264", "test_rec.yuv"):
        os.remove(d / f) if os.path.exists(d / f) else None
    return d


def _run(exe, rundir, args, env=None):
    for f in ("test.264", "test_rec.yuv"):
        if os.path.exists(rundir / f):
            os.remove(rundir / f)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([os.path.join(RDIR, exe)] + args, cwd=rundir, env=e,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    out = []
    for f in ("test.264", "test_rec.yuv"):
        with open(rundir / f, "rb") as fh:
            out.append(hashlib.md5(fh.read()).hexdigest())
    return out


def test_known_answer_with_oracle_swapped_in(rundir):
    """default encoder.cfg: the reference ships the expected bitstream and reconstruction."""
    _run("jm_swap", rundir, [])
    for f in ("test.264", "test_rec.yuv"):
        with open(rundir / f, "rb") as a, open(os.path.join(REF, "bin", f), "rb") as b:
            assert a.read() == b.read(), f


CFGS = ["encoder.cfg", "encoder_baseline.cfg", "encoder_main.cfg", "encoder_extended.cfg", "encoder_yuv422.cfg"]
MATRIX = [(c, ["-d", c, "-p", "SearchMode=%d" % sm]) for c in CFGS for sm in (-1, 0, 1, 2, 3)]
VARIANTS = {
    "baseline FS R16": "-d encoder_baseline.cfg -p SearchMode=-1 -p SearchRange=16",
    "baseline FS rdopt0": "-d encoder_baseline.cfg -p SearchMode=-1 -p RDOptimization=0",
    "baseline FFS rdopt0": "-d encoder_baseline.cfg -p SearchMode=0 -p RDOptimization=0",
    "main FS rdopt0": "-d encoder_main.cfg -p SearchMode=-1 -p RDOptimization=0",
    "main FFS ChromaME1": "-d encoder_main.cfg -p SearchMode=0 -p ChromaMEEnable=1",
    "main FS ChromaME1": "-d encoder_main.cfg -p SearchMode=-1 -p ChromaMEEnable=1",
    "main FS ChromaME2": "-d encoder_main.cfg -p SearchMode=-1 -p ChromaMEEnable=2",
    "422 FS ChromaME2": "-d encoder_yuv422.cfg -p SearchMode=-1 -p ChromaMEEnable=2",
    "422 FFS ChromaME1": "-d encoder_yuv422.cfg -p SearchMode=0 -p ChromaMEEnable=1",
    "main FS WP": "-d encoder_main.cfg -p SearchMode=-1 -p WeightedPrediction=1 -p WeightedBiprediction=1 -p UseWeightedReferenceME=1",
    "main FFS WP": "-d encoder_main.cfg -p SearchMode=0 -p WeightedPrediction=1 -p WeightedBiprediction=1 -p UseWeightedReferenceME=1",
    "main FS WBP implicit": "-d encoder_main.cfg -p SearchMode=-1 -p WeightedBiprediction=2",
    "main FFS bipred SAD-all R8": "-d encoder_main.cfg -p SearchMode=0 -p MEDistortionHPel=0 -p MEDistortionQPel=0 -p BiPredMESearchRange=8 -p BiPredMERefinements=1",
    "high 8x8 FS bipred subpel1": "-d encoder.cfg -p SearchMode=-1 -p BiPredMESubPel=1",
    "high rdopt0 8x8 CAVLC": "-d encoder.cfg -p SearchMode=0 -p RDOptimization=0 -p SymbolMode=0 -p Transform8x8Mode=1",
    "high rdopt0 8x8 SAD-md": "-d encoder.cfg -p SearchMode=-1 -p RDOptimization=0 -p MDDistortion=0 -p Transform8x8Mode=1",
    "main EPZS WP": "-d encoder_main.cfg -p SearchMode=3 -p WeightedPrediction=1 -p UseWeightedReferenceME=1",
    "main FS SATD-fpel": "-d encoder_main.cfg -p SearchMode=-1 -p MEDistortionFPel=2",
    "main FS SAD-all": "-d encoder_main.cfg -p SearchMode=-1 -p MEDistortionHPel=0 -p MEDistortionQPel=0",
    "high 8x8 FS": "-d encoder.cfg -p SearchMode=-1 -p Transform8x8Mode=2",
    "high CAVLC 8x8": "-d encoder.cfg -p SymbolMode=0 -p Transform8x8Mode=1",
    "main qp8": "-d encoder_main.cfg -p QPISlice=8 -p QPPSlice=8 -p QPBSlice=9",
    "main qp45": "-d encoder_main.cfg -p QPISlice=45 -p QPPSlice=45 -p QPBSlice=46",
    "main field": "-d encoder_main.cfg -p PicInterlace=1 -p ReferenceReorder=0 -p PocMemoryManagement=0",
    "main mbaff": "-d encoder_main.cfg -p MbInterlace=1 -p ReferenceReorder=0 -p PocMemoryManagement=0",
    "main slices deblock idc2 offsets": "-d encoder_main.cfg -p SearchMode=0 -p SliceMode=1 -p SliceArgument=27 -p LoopFilterParametersFlag=1 -p LoopFilterDisable=2 -p LoopFilterAlphaC0Offset=3 -p LoopFilterBetaOffset=-2",
    "baseline slices deblock across": "-d encoder_baseline.cfg -p SearchMode=0 -p SliceMode=1 -p SliceArgument=40 -p LoopFilterParametersFlag=1 -p LoopFilterAlphaC0Offset=-3 -p LoopFilterBetaOffset=4",
    "high 8x8 deblock qp40": "-d encoder.cfg -p SearchMode=0 -p Transform8x8Mode=1 -p QPISlice=40 -p QPPSlice=40 -p QPBSlice=41",
    "422 deblock qp38 8x8": "-d encoder_yuv422.cfg -p SearchMode=0 -p Transform8x8Mode=1 -p QPISlice=38 -p QPPSlice=38 -p QPBSlice=39",
    # EPZS (oracle/jmo_epzs.c) and UMHexagonS (oracle/jmo_umhex.c) walkers swapped in: more than the cfg x SearchMode rows above
    "epzs 422": "-d encoder_yuv422.cfg -p SearchMode=3 -p WeightedPrediction=1 -p UseWeightedReferenceME=1",
    "epzs SATD-fpel 8x8 (config 3)": "-d encoder.cfg -p SearchMode=3 -p MEDistortionFPel=2 -p Transform8x8Mode=1",
    "epzs rdopt0": "-d encoder_main.cfg -p SearchMode=3 -p RDOptimization=0",
    "epzs patterns sb/pmvfast fixed1": "-d encoder_main.cfg -p SearchMode=3 -p EPZSPattern=4 -p EPZSDualRefinement=6 -p EPZSFixedPredictors=1",
    "epzs no temporal no spatial-mem square": "-d encoder_main.cfg -p SearchMode=3 -p EPZSTemporal=0 -p EPZSSpatialMem=0 -p EPZSPattern=1 -p EPZSDualRefinement=0",
    "epzs ChromaME1": "-d encoder_main.cfg -p SearchMode=3 -p ChromaMEEnable=1",
    "epzs WBP implicit": "-d encoder_main.cfg -p SearchMode=3 -p WeightedBiprediction=2",
    "epzs R32 3 refs": "-d encoder_baseline.cfg -p SearchMode=3 -p SearchRange=32 -p NumberReferenceFrames=3",
    "umhex 422 WP (config 5)": "-d encoder_yuv422.cfg -p SearchMode=1 -p WeightedPrediction=1 -p UseWeightedReferenceME=1",
    "umhex rdopt0": "-d encoder_main.cfg -p SearchMode=1 -p RDOptimization=0",
    "umhex no DSR R32": "-d encoder_main.cfg -p SearchMode=1 -p UMHexDSR=0 -p SearchRange=32",
    "umhex 3 refs restricted range": "-d encoder_baseline.cfg -p SearchMode=1 -p NumberReferenceFrames=3 -p RestrictSearchRange=0",
    "umhex SATD-fpel 8x8": "-d encoder.cfg -p SearchMode=1 -p MEDistortionFPel=2 -p Transform8x8Mode=1",
    "umhex WBP implicit qp40": "-d encoder_main.cfg -p SearchMode=1 -p WeightedBiprediction=2 -p QPPSlice=40 -p QPISlice=40",
    "umhex scale0 qp20": "-d encoder_main.cfg -p SearchMode=1 -p UMHexScale=0 -p QPPSlice=20 -p QPISlice=20 -p QPBSlice=22",
    "444": "-d encoder_yuv422.cfg -p YUVFormat=3 -p ProfileIDC=244 -p InputFile=foreman_part_qcif_444.yuv",
}


@pytest.mark.parametrize("name,args", [("%s sm%s" % (c, a[-1].split("=")[1]), a) for c, a in MATRIX]
                         + [(k, v.split()) for k, v in VARIANTS.items()])
def test_bitstream_identical_with_oracle_swapped_in(rundir, name, args):
    plain = _run("jm_plain", rundir, args)
    swapped = _run("jm_swap", rundir, args)
    assert plain == swapped, name


@pytest.mark.parametrize("args,names", [
    ("-d encoder_main.cfg -p SearchMode=3", ["EPZSPelBlockMotionSearch", "EPZSSubPelBlockMotionSearch", "EPZSBiPredBlockMotionSearch", "EPZSSubPelBlockSearchBiPred"]),
    ("-d encoder_main.cfg -p SearchMode=1", ["UMHEXSetMotionVectorPredictor", "UMHEXIntegerPelBlockMotionSearch", "UMHEXSubPelBlockMotionSearch",
                                             "UMHEXBipredIntegerPelBlockMotionSearch"]),
])
def test_walkers_are_answered_by_the_oracle(rundir, args, names):
    """The byte-identical rows above would also pass if the harness silently forwarded to JM: every walker call must be SERVED."""
    import re
    e = dict(os.environ, JMO_SWAP_STATS="1")
    r = subprocess.run([os.path.join(RDIR, "jm_swap")] + args.split(), cwd=rundir, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode()
    for n in names:
        m = re.search(r"%s\s+(\d+) served (\d+)" % n, err)
        assert m and int(m.group(1)) > 0 and m.group(1) == m.group(2), (n, m and m.groups())
