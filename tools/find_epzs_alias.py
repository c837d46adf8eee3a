"""Development aid: runs the ORACLE's EPZS slice search over a synthetic clip and reports, per picture, how many integer searches ran and how
many map tests were answered by an aliased 16-bit stamp (oracle/jmo_epzs.c map_shadow) -- used to pick the clip of
tests/test_slice_gpu.py::test_epzs_visited_map_aliases. CPU only."""
import argparse
import ctypes as C
import sys, os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle
from tests.conftest import load_pkg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="720x576")
    ap.add_argument("--range", type=int, default=32)
    ap.add_argument("--refs", type=int, default=1)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--seeds", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    for seed in range(a.seed, a.seed + a.seeds):
        a.seed_now = seed
        fields = [run(a, ideal) for ideal in (0, 1)]
        diff = [k for k, (x, y) in enumerate(zip(*fields)) if x[0] != y[0]]
        print("seed %d: alias events after each picture %s; pictures whose search records differ from what a per-search map would give: %s" % (seed, [x[2] for x in fields[0]], diff), flush=True)


def run(a, ideal):
    W, H = [int(v) for v in a.size.split("x")]
    pkg = load_pkg()
    rng = np.random.default_rng(a.seed_now)
    nref, R = a.refs, a.range
    clip = pkg.slice_host.synth_clip(rng, W, H, a.frames + nref - 1)
    qp = 28
    lam = int(65536 * np.sqrt(0.85 * 2 ** ((qp - 12) / 3.0)) + 0.5)
    ref_cost1 = int(2 * np.sqrt(0.85 * 2 ** ((qp - 12) / 3.0)))
    nmb = (W // 16) * (H // 16)
    epzs = oracle.Epzs(W, H, R, nref)
    oracle.lib().jmo_epzs_ideal_map.argtypes = [C.c_void_p, C.c_int]
    oracle.lib().jmo_epzs_ideal_map(epzs.h, ideal)
    out = []
    all_mv_state = np.zeros((4, 4, oracle.MAX_REFS, 9, 2), np.int16)
    prev_field = [(np.zeros((H // 4, W // 4, 2), np.int16), np.full((H // 4, W // 4), -1, np.int64))] * 2
    for f in range(nref, nref + a.frames - 1):
        cur = clip[f]
        refs = [clip[f - 1 - r] for r in range(nref)]
        pocs = [2 * (f - 1 - r) for r in range(nref)]
        orefs = [oracle.RefPic(r, yuv_format=0) for r in refs]
        epzs.slice_init(2 * f, pocs, pocs, [prev_field[0][0], prev_field[1][0]], [prev_field[0][1], prev_field[1][1]])
        ref_idx = np.full((H // 4, W // 4), -1, np.int8)
        mvf = np.zeros((H // 4, W // 4, 2), np.int16)
        q = oracle.lowcplx_params(3, R, nref, [lam] * 3, ref_cost1, W, H, epzs=epzs, umhex=None, all_mv_state=all_mv_state)
        sid = np.zeros(nmb, np.int32)
        q._sid = sid
        q.slice_id = sid.ctypes.data
        want, _, _ = oracle.lowcplx_p_slice(q, orefs, cur, ref_idx, mvf, mb_first=0, mb_count=nmb)
        if not ideal and a.verbose:
            print("picture %d: %d searches so far, %d alias events so far" % (f, epzs.search_count(), epzs.alias_events()), flush=True)
        out.append((want.tobytes(), mvf.copy(), epzs.alias_events()))
        ids = np.where(ref_idx >= 0, np.array(pocs, np.int64)[np.clip(ref_idx, 0, None)], -1)
        prev_field = [(mvf.copy(), ids), prev_field[0]]
    return out


if __name__ == "__main__":
    main()
