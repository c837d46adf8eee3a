#!/usr/bin/env python3
"""Times jmhip_p_slice_search (the macroblock-wavefront P-slice kernel) on a synthetic 1080p picture for each search mode.
usage: python tools/time_slice.py [--size 1080p|qcif|720p] [--refs N] [--modes 3,1,0,-1] [--range R] [--frames F]
Every timed call searches a NEW picture of a moving synthetic clip against the previous `refs` source frames (the first call starts from an
empty state, the later ones from what the previous picture left -- the relaxation schedule's first guess); sweeps / passes per call are printed."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1080p")
    ap.add_argument("--refs", type=int, default=1)
    ap.add_argument("--modes", default="3,1,0,-1")
    ap.add_argument("--range", type=int, default=32)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--slices", type=int, default=1, help="fixed-size slices per picture (search modes -1, 0, 2)")
    ap.add_argument("--clip", choices=["smooth", "bench"], default="smooth", help="bench: bench.py's 1080p translation + noise clip (slower to settle)")
    ap.add_argument("--rdopt", type=int, default=0, help="the call records of RDOptimization 1 / 2 (search modes -1, 0, 2; the speculative binding's call)")
    ap.add_argument("--download", action="store_true", help="include the download of the result records in the timed region")
    ap.add_argument("--per-slice-calls", action="store_true", help="with --slices: one call per slice instead of slice_mbs")
    a = ap.parse_args()
    W, H = {"1080p": (1920, 1088), "720p": (1280, 720), "qcif": (176, 144), "2160p": (3840, 2160)}[a.size]
    pkg = ge._load_pkg()
    lib = pkg.load_library()
    rng = np.random.default_rng(3)
    nfr = a.frames
    clip = pkg.slice_host.synth_clip(rng, W, H, a.refs + nfr)
    if a.clip == "bench":
        import bench
        assert (W, H) == (bench.W, bench.H), "--clip bench is the 1080p clip"
        fr = bench.synth_frames(4, False)
        clip = [fr[i % 4][0] for i in range(a.refs + nfr)]
    ctx = pkg.Context(W, H, yuv_format=0, max_refs=a.refs, search_range=a.range)
    lam = int(65536 * np.sqrt(0.85 * 2 ** ((28 - 12) / 3.0)) + 0.5)
    for mode in [int(m) for m in a.modes.split(",")]:
        ctx.slice_state_reset()
        ts, sw = [], []
        for f in range(a.refs, a.refs + nfr):
            for r in range(a.refs):
                ctx.ref_upload(r, clip[f - 1 - r])
                ctx.interp_luma(r)
            ctx.cur_upload(clip[f])
            ctx.epzs_colocated_upload(np.zeros((H // 4, W // 4, 2), np.int16))
            p = pkg.slice_host.slice_params(mode, a.range, a.refs, [lam] * 3, 10, W, H=H)
            pocs = [2 * (f - 1 - r) for r in range(a.refs)]
            lib.jmhip_epzs_scales(p, 2 * f, (C.c_int * a.refs)(*pocs), a.refs)
            nmb = (W // 16) * (H // 16)
            per = (nmb + a.slices - 1) // a.slices
            ctx.sync()
            t0 = time.perf_counter()
            if a.slices > 1 and a.per_slice_calls:
                tot = 0
                for k in range(a.slices):
                    p.mb_first, p.mb_count = k * per, min(per, nmb - k * per)
                    ctx.p_slice_search(p, download=False)
                    tot += ctx.slice_passes()
                ctx.sync()
                ts.append(time.perf_counter() - t0)
                sw.append(tot)
                continue
            if a.slices > 1:
                p.slice_mbs = per
            p.rdopt = a.rdopt
            ctx.p_slice_search(p, download=a.download)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
            sw.append(ctx.slice_passes())
        print("mode %2d  %s  refs %d  R %d:  %s ms per picture  (sweeps/passes %s)  -> %.0f macroblocks/s (mean of pictures 2..)" % (
            mode, a.size, a.refs, a.range, " ".join("%.1f" % (t * 1e3) for t in ts), " ".join(str(v) for v in sw),
            (W // 16) * (H // 16) / (sum(ts[1:]) / max(1, len(ts) - 1))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
