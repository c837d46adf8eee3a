"""Host-side arithmetic of the slice search (no GPU): the search-mode setup helpers of libjmhip.so reproduce JM's initialisation
(EPZSInit thresholds and window predictors, the POC-distance scales of EPZSSliceInit, the float thresholds of UMHEX_DefineThreshold)
exactly as the pinned oracle does."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle


@pytest.mark.parametrize("qp,scale,width", [(28, 3, 176), (28, 3, 1920), (20, 0, 352), (40, 3, 3840), (51, 1, 720), (0, 3, 176)])
def test_umhex_thresholds_match_the_oracle(pkg, qp, scale, width):
    lib = pkg.load_library()
    p = pkg.SliceParams()
    lib.jmhip_umhex_setup(p, 1, scale, qp, width)
    u = oracle.Umhex(width, 144, 16, 2, qp, scale=scale)
    med, big, multi, dsr, bsize, a1, a2 = oracle.umhex_thresholds(u)
    u.close()
    for bt in range(1, 8):
        assert (p.umhex_thres[0][bt], p.umhex_thres[1][bt], p.umhex_thres[2][bt], p.umhex_thres[3][bt]) == (med[bt], big[bt], multi[bt], dsr[bt])
        assert np.float32(p.umhex_bsize[bt]).tobytes() == np.float32(bsize[bt]).tobytes()
        assert np.float32(p.umhex_alpha1[bt]).tobytes() == np.float32(a1[bt]).tobytes() and np.float32(p.umhex_alpha2[bt]).tobytes() == np.float32(a2[bt]).tobytes()


@pytest.mark.parametrize("R", [8, 16, 32])
def test_epzs_setup_matches_the_oracle(pkg, R):
    lib = pkg.load_library()
    p = pkg.SliceParams()
    lib.jmhip_epzs_setup(p, R, 2, 3, 2, 1, 1, 1, 0, 1, 2, 2)
    e = oracle.Epzs(176, 144, R, 2)
    for which in range(4):
        for bt in range(1, 8):
            assert p.epzs_thres[which][bt] == oracle.epzs_threshold(e, which, bt)
    e.close()
    levels = {8: 2, 16: 3, 32: 4}[R]                      # RoundLog2(R) - 1 (me_epzs.c:339)
    assert (p.epzs_nwin, p.epzs_nwin_ext) == (8 * levels, 20 * levels)


def test_epzs_scales_match_the_oracle(pkg):
    lib = pkg.load_library()
    L = oracle._walker_protos()
    L.jmo_epzs_mv_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    for poc, pocs in ((8, [6, 4, 2, 0]), (4, [2, 0]), (2, [0]), (20, [18, 10, 2])):
        p = pkg.SliceParams()
        lib.jmhip_epzs_scales(p, poc, (C.c_int * len(pocs))(*pocs), len(pocs))
        e = oracle.Epzs(176, 144, 16, 4, temporal=0)
        z = np.zeros((36, 44, 2), np.int16)
        zi = np.zeros((36, 44), np.int64)
        e.slice_init(poc, pocs, list(range(len(pocs))), [z, z], [zi, zi])
        for i in range(len(pocs)):
            for k in range(len(pocs)):
                assert p.epzs_mv_scale[i][k] == L.jmo_epzs_mv_scale(e.h, 0, i, k), (poc, pocs, i, k)
        e.close()
