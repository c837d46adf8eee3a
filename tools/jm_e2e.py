#!/usr/bin/env python3
"""bench.py's end-to-end legs alone (the real JM, plain and bound to libjmhip.so, on the 1080p bench clip): wall time, JM's own P-frame
timers, the shim's per-hook wall time (JMHIP_SHIM_STATS) and the byte comparison of the bitstreams.  usage: tools/jm_e2e.py [rdopt0|config3|rdopt1 ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    which = sys.argv[1:] or ["rdopt0", "config3", "rdopt1"]
    frames = bench.synth_frames(4, False)
    for w in which:
        r = bench.jm_end_to_end(frames, config3=(w == "config3"), rdopt1=(w == "rdopt1"))
        print(w, json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
