#!/usr/bin/env python3
"""Generates the frame-level golden fixtures of the low-complexity P-slice inter decision from the REAL JM (container only).

Runs oracle/_ref/jm_field (the unmodified reference encoder with recording interposers, oracle/tap/tap_field.c) on the reference's own
QCIF clip with RDOptimization=0, DisableIntraInInter=1 and packs, per coded P picture: source luma, reference lumas, slice constants,
every BlockMotionSearch call (predictor, vector, cost) and the final vector field. Only data leaves the reference.

    make -C oracle ref && python tests/golden/make_golden_field.py
"""
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/bin"
TAP = os.path.join(ROOT, "oracle", "_ref", "jm_field")
OUT = os.path.dirname(os.path.abspath(__file__))
COMMON = ["-d", "encoder_baseline.cfg", "-p", "RDOptimization=0", "-p", "DisableIntraInInter=1", "-p", "FramesToBeEncoded=4", "-p", "SearchRange=16"]
RUNS = {
    "field_full_r16_1ref": COMMON + ["-p", "SearchMode=-1", "-p", "NumberReferenceFrames=1"],
    "field_fastfull_r16_2ref": COMMON + ["-p", "SearchMode=0", "-p", "NumberReferenceFrames=2"],
    "field_epzs_r16_2ref": COMMON + ["-p", "SearchMode=3", "-p", "NumberReferenceFrames=2"],
    "field_umhex_r16_2ref": COMMON + ["-p", "SearchMode=1", "-p", "NumberReferenceFrames=2"],
    "field_umhexsmp_r16_2ref": COMMON + ["-p", "SearchMode=2", "-p", "NumberReferenceFrames=2"],      # the simplified UMHexagonS (me_umhexsmp.c)
    "field_umhexsmp_t8_q34_r16_2ref": COMMON + ["-p", "SearchMode=2", "-p", "NumberReferenceFrames=2", "-p", "Transform8x8Mode=1", "-p", "ProfileIDC=100",
                                                "-p", "AdaptiveRounding=0", "-p", "QPPSlice=34", "-p", "MEDistortionFPel=2"],
    # Transform8x8Mode (High profile, AdaptiveRounding off): 1 = both transform sizes compete in every mode and the P8x8 partitioning depends on the
    # coded-block pattern of the 8x8-transform pass (coarse quantiser: the pattern is often empty); 2 = 8x8 only
    "field_epzs_t8_r16_2ref": COMMON + ["-p", "SearchMode=3", "-p", "NumberReferenceFrames=2", "-p", "Transform8x8Mode=1", "-p", "ProfileIDC=100",
                                        "-p", "AdaptiveRounding=0", "-p", "QPPSlice=36"],
    "field_full_t8only_r16_1ref": COMMON + ["-p", "SearchMode=-1", "-p", "NumberReferenceFrames=1", "-p", "Transform8x8Mode=2", "-p", "ProfileIDC=100",
                                            "-p", "AdaptiveRounding=0"],
    "field_umhex_t8_cabac_r16_2ref": COMMON + ["-p", "SearchMode=1", "-p", "NumberReferenceFrames=2", "-p", "Transform8x8Mode=1", "-p", "ProfileIDC=100",
                                               "-p", "AdaptiveRounding=0", "-p", "QPPSlice=32", "-p", "SymbolMode=1"],
    # SSE (computeSSE, me_distortion.c:1042) in the refinements and as the mode-decision metric, with both transform sizes; EPZS with SSE at every level
    "field_full_sse_t8_r16_1ref": COMMON + ["-p", "SearchMode=-1", "-p", "NumberReferenceFrames=1", "-p", "Transform8x8Mode=1", "-p", "ProfileIDC=100",
                                            "-p", "AdaptiveRounding=0", "-p", "QPPSlice=32", "-p", "MEDistortionHPel=1", "-p", "MEDistortionQPel=1", "-p", "MDDistortion=1"],
    "field_epzs_sse_r16_2ref": COMMON + ["-p", "SearchMode=3", "-p", "NumberReferenceFrames=2", "-p", "MEDistortionFPel=1", "-p", "MEDistortionHPel=1",
                                         "-p", "MEDistortionQPel=1", "-p", "MDDistortion=1"],
    "field_fastfull_t8_r16_2ref": COMMON + ["-p", "SearchMode=0", "-p", "NumberReferenceFrames=2", "-p", "Transform8x8Mode=1", "-p", "ProfileIDC=100",
                                            "-p", "AdaptiveRounding=0", "-p", "QPPSlice=40"],
}


def parse(path):
    frames = []
    with open(path, "rb") as f:
        data = f.read()
    off = 0
    while off < len(data):
        magic, kind, n = struct.unpack_from("<iii", data, off)
        assert magic == 0x4a4d5450
        off += 12
        r = np.frombuffer(data, dtype="<i4", count=n, offset=off).copy()
        off += 4 * n
        if kind == 20:
            frames.append({"frame": r})
        elif kind == 21:
            frames[-1]["calls"] = r.reshape(-1, 12)
        else:
            frames[-1]["field"] = r
    return frames


def main():
    if not os.path.exists(TAP):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    only = sys.argv[1:]
    for name, args in RUNS.items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as d:
            for f in os.listdir(REF):
                if f.endswith(".cfg") or f.endswith(".yuv"):
                    shutil.copy(os.path.join(REF, f), d)
            env = dict(os.environ, JM_TAP_OUT=os.path.join(d, "tap.bin"))
            subprocess.run([TAP] + args, cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            frames = parse(os.path.join(d, "tap.bin"))
        out = {"n_frames": np.int32(len(frames))}
        for k, fr in enumerate(frames):
            h = fr["frame"]
            W, H, qp, nref = (int(v) for v in h[:4])
            w4, h4, nmb = W // 4, H // 4, (W // 16) * (H // 16)
            out["f%d_head" % k] = h[:12].astype(np.int32)          # W H qp nrefs img_number frame_ctr_b lambda_mf[3] ref_cost1 poc num_ref_idx_l0_active
            p = 12
            out["f%d_refinfo" % k] = h[p:p + 3 * nref].reshape(nref, 3).astype(np.int32)   # poc, ref_pic_num lo, hi
            p += 3 * nref
            out["f%d_cur" % k] = h[p:p + W * H].reshape(H, W).astype(np.uint8)
            p += W * H
            out["f%d_refs" % k] = h[p:p + nref * W * H].reshape(nref, H, W).astype(np.uint8)
            p += nref * W * H
            col = h[p:p + 2 * w4 * h4 * 4].reshape(2, h4, w4, 4)
            out["f%d_col_mv" % k] = col[..., :2].astype(np.int16)
            out["f%d_col_ref_id" % k] = (col[..., 2].astype(np.int64) & 0xffffffff) | (col[..., 3].astype(np.int64) << 32)
            out["f%d_calls" % k] = fr["calls"].astype(np.int32)
            fl = fr["field"]
            out["f%d_mb" % k] = fl[:6 * nmb].reshape(nmb, 6).astype(np.int16)              # mb_type, slice_nr, b8mode[4]
            out["f%d_field" % k] = fl[6 * nmb:].reshape(h4, w4, 3).astype(np.int16)        # ref_idx, mvx, mvy
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, len(frames), "P frames, %d calls in the first" % len(frames[0]["calls"]), "%.0f KB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))


if __name__ == "__main__":
    main()
