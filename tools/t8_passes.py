"""Sweeps of the relaxation schedule with and without Transform8x8Mode (diagnostic): python tools/t8_passes.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from tests.test_slice_gpu import run_synthetic
pkg = ge._load_pkg()
for mode in (3, -1, 0, 1):
    for t8, qp in ((0, 36), (1, 36), (2, 36)):
        print("mode", mode, "t8", t8, "passes", run_synthetic(pkg, mode, 320, 192, 16, 2, t8=t8, qp=qp), flush=True)
