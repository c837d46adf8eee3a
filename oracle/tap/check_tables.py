#!/usr/bin/env python3
"""Container-only check (needs /root/reference): the oracle's 8x8 quantiser tables, which oracle/jmo_tq.c builds from six
position classes, against the literal tables of the reference (lencod/src/transform8x8.c:39-167, read as text and parsed;
nothing is copied into the repo). Also the 4x4 tables (block.c:39-55). Run: python oracle/tap/check_tables.py"""
import ctypes
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("JMROOT", "/root/reference")


def parse_table(text, name):
    m = re.search(r"\b%s\s*(\[[^\]]*\])+\s*=\s*\{" % re.escape(name), text)
    if not m:
        raise SystemExit("table %s not found" % name)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return [int(v) for v in re.findall(r"-?\d+", re.sub(r"/\*.*?\*/|//[^\n]*", "", text[m.end():i], flags=re.S))]


def main():
    lib = ctypes.CDLL(os.path.join(HERE, "..", "liboracle.so"))
    t8 = open(os.path.join(REF, "lencod/src/transform8x8.c")).read()
    blk = open(os.path.join(REF, "lencod/src/block.c")).read()
    bad = 0
    for name, fn in (("quant_coef8", lib.jmo_quant_coef8), ("dequant_coef8", lib.jmo_dequant_coef8)):
        ref = parse_table(t8, name)
        assert len(ref) == 6 * 64, (name, len(ref))
        for k in range(6):
            for j in range(8):
                for i in range(8):
                    if fn(k, j, i) != ref[k * 64 + j * 8 + i]:
                        bad += 1
    for name, sym in (("quant_coef", "jmo_quant_coef"), ("dequant_coef", "jmo_dequant_coef")):
        ref = parse_table(blk, name)
        assert len(ref) == 6 * 16, (name, len(ref))
        arr = (ctypes.c_int * 96).in_dll(lib, sym)
        bad += sum(1 for a, b in zip(arr, ref) if a != b)
    print("check_tables: %s" % ("OK (4 tables, 960 entries)" if not bad else "%d MISMATCHES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
