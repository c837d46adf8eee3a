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
