#!/usr/bin/env python3
"""Generates the committed golden fixtures from the REAL JM (container only).

Runs oracle/_ref/jm_tap (the unmodified reference encoder with recording interposers, oracle/tap/tap_capture.c) on the
reference's own clips and configs, parses the int32 record stream and packs a sample of records per kind into
tests/golden/*.npz. Only data leaves the reference: inputs and expected outputs of individual calls.

    make -C oracle ref && python tests/golden/make_golden.py
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
TAP = os.path.join(ROOT, "oracle", "_ref", "jm_tap")
OUT = os.path.dirname(os.path.abspath(__file__))

RUNS = {
    # name: (args, max records per kind kept)
    "default_high_fastfull": ([], 60),                                                                        # bin/encoder.cfg: High, CABAC, 8x8, FastFull +-32
    "baseline_fs16": (["-d", "encoder_baseline.cfg", "-p", "SearchMode=-1", "-p", "SearchRange=16"], 60),   # BASELINE.json configs[0]
    "yuv422_fs": (["-d", "encoder_yuv422.cfg", "-p", "SearchMode=-1", "-p", "SearchRange=16"], 40),
    "main_cavlc8x8_rdopt0": (["-d", "encoder_main.cfg", "-p", "SearchMode=0", "-p", "RDOptimization=0", "-p", "SymbolMode=0",
                              "-p", "Transform8x8Mode=1", "-p", "ProfileIDC=100"], 40),
    # the other error metrics of computeUniPred (mv-search.c:400-424): Hadamard SAD at every level (the refinements start from the carried
    # minimum), SSE at integer and half-pel positions with SAD quarter-pel; FastFullSearch with a non-SAD metric (squared error surfaces)
    "baseline_fs16_satd": (["-d", "encoder_baseline.cfg", "-p", "SearchMode=-1", "-p", "SearchRange=16", "-p", "MEDistortionFPel=2",
                            "-p", "MEDistortionHPel=2", "-p", "MEDistortionQPel=2", "-p", "FramesToBeEncoded=2"], 40),
    "baseline_fs16_sse": (["-d", "encoder_baseline.cfg", "-p", "SearchMode=-1", "-p", "SearchRange=16", "-p", "MEDistortionFPel=1",
                           "-p", "MEDistortionHPel=1", "-p", "MEDistortionQPel=0", "-p", "FramesToBeEncoded=2"], 40),
    "main_fastfull_satd_fpel": (["-d", "encoder_main.cfg", "-p", "SearchMode=0", "-p", "MEDistortionFPel=2", "-p", "MEDistortionHPel=2",
                                 "-p", "MEDistortionQPel=2", "-p", "FramesToBeEncoded=2", "-p", "NumberBFrames=0"], 40),
}
KINDS = {1: "luma", 2: "chroma", 3: "fullpel", 4: "subpel", 5: "fastfull", 6: "dct4", 7: "dct8", 8: "dct16", 9: "dctc"}


def parse(path):
    """Stream order matters: a search record's first field (JM picture id = StorablePicture address slot) is replaced by
    the index of the most recent luma record of that picture, because JM recycles StorablePicture addresses."""
    recs = {k: [] for k in KINDS.values()}
    with open(path, "rb") as f:
        data = f.read()
    off = 0
    latest = {}
    while off < len(data):
        magic, kind, n = struct.unpack_from("<iii", data, off)
        assert magic == 0x4a4d5450
        off += 12
        r = np.frombuffer(data, dtype="<i4", count=n, offset=off).copy()
        off += 4 * n
        name = KINDS[kind]
        if name == "luma":
            latest[int(r[0])] = len(recs["luma"])
        elif name in ("fullpel", "subpel", "fastfull"):
            r[0] = latest[int(r[0])]
        recs[name].append(r)
    return recs


def subsample(lst, n):
    if len(lst) <= n:
        return lst
    idx = np.linspace(0, len(lst) - 1, n).astype(int)
    return [lst[i] for i in idx]


def main():
    if not os.path.exists(TAP):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    only = sys.argv[1:]                                # optional: the runs to (re)generate; default all
    for name, (args, keep) in RUNS.items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as d:
            for f in os.listdir(REF):
                if f.endswith(".cfg") or f.endswith(".yuv"):
                    shutil.copy(os.path.join(REF, f), d)
            env = dict(os.environ, JM_TAP_OUT=os.path.join(d, "tap.bin"))
            subprocess.run([TAP] + args, cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            recs = parse(os.path.join(d, "tap.bin"))
        out = {}
        for kind, lst in recs.items():
            if not lst:
                continue
            if kind not in ("luma", "chroma"):          # reference pictures are all kept: the search records replay on them
                lst = subsample(lst, keep)
            L = max(len(r) for r in lst)
            arr = np.zeros((len(lst), L), dtype=np.int32)
            lens = np.zeros(len(lst), dtype=np.int32)
            for i, r in enumerate(lst):
                arr[i, :len(r)] = r
                lens[i] = len(r)
            # samples, levels and tables are small numbers: store as the narrowest exact dtype
            out[kind] = arr
            out[kind + "_len"] = lens
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items() if not k.endswith("_len")},
              "%.0f KB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))


if __name__ == "__main__":
    main()
