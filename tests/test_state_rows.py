"""The stored-row layout of the slice search's EPZS row memories (me_wave.hip: state_row / mb_stage / mb_commit / epzs_rows_fold_kernel),
restated on the host: under ANY evaluation order that ends at the fixpoint, a macroblock must read, per 4x4 column, what JM's single-row
array would hold when JM reaches that macroblock in coding order. The test drives the host mirror with random slices (also beginning and
ending mid-row) and compares with a literal single-row simulation -- the device code follows the same three rules."""
import numpy as np
import pytest


def state_row(col, mbx, mby, mbw, first, count, row0):
    """me_wave.hip state_row: the stored row (0 = what the slice found) that holds column `col` for macroblock (mbx, mby)."""
    cx = col >> 2
    yy = mby if cx < mbx else mby - 1
    if yy < 0:
        return 0
    a = yy * mbw + cx
    if a < first or a >= first + count:
        return 0
    return yy - row0 + 1


def fold(rows, mbw, first, count, row0):
    """epzs_rows_fold_kernel: row 0 after the slice = JM's single-row array after the slice."""
    out = rows[0].copy()
    last = first + count - 1
    ylast = last // mbw
    for col in range(4 * mbw):
        cx = col >> 2
        for yy in (ylast, ylast - 1):
            if yy < 0:
                continue
            a = yy * mbw + cx
            if first <= a <= last:
                out[col] = rows[yy - row0 + 1][col]
                break
    return out


@pytest.mark.parametrize("seed", range(12))
def test_stored_rows_equal_the_single_row_array(seed):
    rng = np.random.default_rng(seed)
    mbw, mbh = int(rng.integers(3, 9)), int(rng.integers(2, 7))
    nmb = mbw * mbh
    single = rng.integers(0, 1000, 4 * mbw)                  # JM's array before the picture
    incoming = single.copy()
    first = 0
    while first < nmb:
        count = int(rng.integers(1, nmb - first + 1))
        row0 = first // mbw
        rows = np.zeros((mbh + 1, 4 * mbw), np.int64)
        rows[0] = incoming
        # what each macroblock writes into its four columns (a function of what it reads, so that order matters)
        def evaluate(addr, read):
            mbx = addr % mbw
            lo, hi = max(0, 4 * mbx - 4), min(4 * mbw, 4 * mbx + 8)
            return [(int(read[lo:hi].sum()) * 31 + addr * 7 + k) % 1000 for k in range(4)]
        # JM: coding order on the single-row array
        for addr in range(first, first + count):
            mbx = addr % mbw
            single[4 * mbx:4 * mbx + 4] = evaluate(addr, single)
        # device: sweeps in a random order until nothing changes, every macroblock reading through state_row
        for sweep in range(4 * nmb):
            changed = False
            for addr in rng.permutation(np.arange(first, first + count)):
                mbx, mby = int(addr) % mbw, int(addr) // mbw
                view = np.array([rows[state_row(c, mbx, mby, mbw, first, count, row0)][c] for c in range(4 * mbw)])
                new = evaluate(int(addr), view)
                r = mby - row0 + 1
                if list(rows[r][4 * mbx:4 * mbx + 4]) != new:
                    rows[r][4 * mbx:4 * mbx + 4] = new
                    changed = True
            if not changed:
                break
        else:
            raise AssertionError("the sweeps did not settle")
        incoming = fold(rows, mbw, first, count, row0)
        assert np.array_equal(incoming, single), "slice [%d, %d) of a %dx%d picture" % (first, first + count, mbw, mbh)
        first += count
