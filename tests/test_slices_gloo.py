"""N > 1 host path on CPU: slice partition + per-frame all-gather of reconstructed bands, world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mbh, w, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tests.conftest import load_pkg
    pkg = load_pkg()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    row0, row1, band = pkg.slices.band_rows(mbh, world, rank)
    rng = np.random.default_rng(99)                      # same picture on every rank
    Y = rng.integers(0, 256, (mbh * 16, w), dtype=np.uint8)
    U = rng.integers(0, 256, (mbh * 8, w // 2), dtype=np.uint8)
    V = rng.integers(0, 256, (mbh * 8, w // 2), dtype=np.uint8)
    bufs = pkg.slices.gather_buffers(torch, world, band, w, 8, w // 2, "cpu")
    views = pkg.slices.send_buffers(torch, band, w, 8, w // 2, "cpu")
    # each rank only "reconstructed" its own band
    views[0][: (row1 - row0) * 16] = torch.from_numpy(Y[row0 * 16:row1 * 16])
    views[1][: (row1 - row0) * 8] = torch.from_numpy(U[row0 * 8:row1 * 8])
    views[2][: (row1 - row0) * 8] = torch.from_numpy(V[row0 * 8:row1 * 8])
    pkg.slices.all_gather_recon(dist, bufs, views)
    ok = (np.array_equal(bufs[0][: mbh * 16].numpy(), Y) and np.array_equal(bufs[1][: mbh * 8].numpy(), U)
          and np.array_equal(bufs[2][: mbh * 8].numpy(), V))
    q.put((rank, ok, row0, row1))
    dist.destroy_process_group()


@pytest.mark.parametrize("mbh", [68, 9])
def test_all_gather_of_slice_bands_rebuilds_the_reference(mbh):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mbh, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert res[0][2] == 0 and res[-1][3] == mbh and res[0][3] == res[1][2]      # bands tile the picture


def test_band_rows_cover_every_macroblock_row_once(pkg):
    for mbh in (68, 135, 9, 1):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for r in range(world):
                a, b, _ = pkg.slices.band_rows(mbh, world, r)
                rows += list(range(a, b))
            assert rows == list(range(mbh))
