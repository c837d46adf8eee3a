"""N > 1 host path on CPU: slice partition + the per-frame one-chunk all-gather of reconstructed bands (what bench.py does), world_size 2 and 3 over gloo."""
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


def test_band_rows_cover_every_macroblock_row_once(pkg):
    for mbh in (68, 135, 9, 1):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for r in range(world):
                a, b, _ = pkg.slices.band_rows(mbh, world, r)
                rows += list(range(a, b))
            assert rows == list(range(mbh))


def _chunk_worker(rank, world, port, mbh, w, q):
    """The exchange bench.py performs per frame, on CPU tensors: pack the own band as ONE chunk [Y | U | V], one all_gather_into_tensor,
    unpack every chunk into the next reference (jmhip_recon_pack_band / jmhip_ref_unpack_bands on the device; their host mirrors here)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tests.conftest import load_pkg
    pkg = load_pkg()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    row0, row1, band = pkg.slices.band_rows(mbh, world, rank)
    rng = np.random.default_rng(7)                       # the "reconstruction": every rank knows it, each sends only its band
    Y = rng.integers(0, 256, (mbh * 16, w), dtype=np.uint8)
    U = rng.integers(0, 256, (mbh * 8, w // 2), dtype=np.uint8)
    V = rng.integers(0, 256, (mbh * 8, w // 2), dtype=np.uint8)
    mine = (np.zeros_like(Y), np.zeros_like(U), np.zeros_like(V))
    mine[0][row0 * 16:row1 * 16], mine[1][row0 * 8:row1 * 8], mine[2][row0 * 8:row1 * 8] = Y[row0 * 16:row1 * 16], U[row0 * 8:row1 * 8], V[row0 * 8:row1 * 8]
    chunk = pkg.slices.chunk_bytes(w, w // 2, 8, band)
    sbuf = torch.from_numpy(pkg.slices.pack_band_host(np, *mine, rank, band, 8))
    gbuf = torch.zeros(chunk * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(gbuf, sbuf)              # the one collective of a frame
    got = pkg.slices.unpack_bands_host(np, gbuf.numpy(), world, band, mbh * 16, w, mbh * 8, w // 2, 8)
    q.put((rank, all(np.array_equal(a, b) for a, b in zip(got, (Y, U, V))), chunk))
    dist.destroy_process_group()


@pytest.mark.parametrize("mbh,world", [(68, 2), (9, 2), (135, 3)])
def test_one_chunk_all_gather_rebuilds_the_reference(mbh, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, mbh, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
