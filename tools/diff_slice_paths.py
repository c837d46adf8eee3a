#!/usr/bin/env python3
"""Development aid: searches the same pictures with jmhip_p_slice_search's two implementations of the exhaustive modes -- the sweeps over the
frame kernels (me_xslice.hip) and the one-wave-per-macroblock kernels (me_wave.hip, JMHIP_SLICE_X=0) -- and prints the first records that differ.
usage: python tools/diff_slice_paths.py [--mode -1|0] [--size WxH] [--range R] [--refs N] [--frames F]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", type=int, default=-1)
    ap.add_argument("--size", default="176x144")
    ap.add_argument("--range", type=int, default=16)
    ap.add_argument("--refs", type=int, default=1)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--rdopt", type=int, default=0)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    pkg = ge._load_pkg()
    rng = np.random.default_rng(5)
    clip = pkg.slice_host.synth_clip(rng, W, H, a.frames + a.refs)
    lam = int(65536 * np.sqrt(0.85 * 2 ** ((28 - 12) / 3.0)) + 0.5)
    ctxs = {}
    for x in ("1", "0"):
        ctxs[x] = pkg.Context(W, H, yuv_format=0, max_refs=a.refs, search_range=a.range)
        ctxs[x].slice_state_reset()
    parts = pkg.partition_table()
    for f in range(a.refs, a.refs + a.frames):
        got = {}
        for x, ctx in ctxs.items():
            os.environ["JMHIP_SLICE_X"] = x
            for r in range(a.refs):
                ctx.ref_upload(r, clip[f - 1 - r])
                ctx.interp_luma(r)
            ctx.cur_upload(clip[f])
            p = pkg.slice_host.slice_params(a.mode, a.range, a.refs, [lam] * 3, 9, W, H=H, rdopt=a.rdopt)
            got[x] = ctx.p_slice_search(p)
            print("frame %d path X=%s: %d sweeps" % (f, x, ctx.slice_passes()))
        g, w = got["1"], got["0"]
        nbad = 0
        order = [0, 1, 2, 3, 4]
        for b8 in range(4):
            order += [5 + b8, 9 + 2 * b8, 10 + 2 * b8, 17 + 2 * b8, 18 + 2 * b8] + [25 + 4 * b8 + k for k in range(4)]
        for i in range(len(g)):
            bad = False
            for r in range(a.refs):           # (JM runs the references inside each partition group; close enough to find the first divergence)
                for pi in order:
                    d = [fld for fld in ("pred", "mv_int", "cost_int", "mv", "cost") if not np.array_equal(g[fld][i][r, pi], w[fld][i][r, pi])]
                    if d:
                        print("  mb %d ref %d partition %d (bt %d at %d,%d): first divergence in %s: sweeps pred %s mv_int %s cost_int %d mv %s cost %d | wave pred %s mv_int %s cost_int %d mv %s cost %d" % (
                            i, r, pi, parts[pi][0], parts[pi][1], parts[pi][2], d, g["pred"][i][r, pi].tolist(), g["mv_int"][i][r, pi].tolist(), g["cost_int"][i][r, pi], g["mv"][i][r, pi].tolist(), g["cost"][i][r, pi],
                            w["pred"][i][r, pi].tolist(), w["mv_int"][i][r, pi].tolist(), w["cost_int"][i][r, pi], w["mv"][i][r, pi].tolist(), w["cost"][i][r, pi]))
                        bad = True
                        break
                if bad:
                    break
            for fld in ("best_mode", "min_cost", "b8mode", "b8ref", "final_mv", "skip_mv"):
                if not np.array_equal(g[fld][i], w[fld][i]):
                    print("  mb %d %s: sweeps %s wave %s" % (i, fld, g[fld][i].tolist(), w[fld][i].tolist()))
                    bad = True
                    break
            nbad += bad
            if nbad >= 6:
                break
        print("frame %d: %s" % (f, "identical" if nbad == 0 else "%d+ differences" % nbad))
    for ctx in ctxs.values():
        ctx.close()


if __name__ == "__main__":
    main()
