"""Slice sharding across ranks (host logic): SURVEY 8(e).

Rank r of `world` owns macroblock rows [r*B, min((r+1)*B, mbh)), B = ceil(mbh / world): JM's SliceMode=1 (FIXED_MB,
inc/global.h:175-181; src/slice.c:214) with SliceArgument a multiple of the macroblock row. Because every band
starts at r*B, an all-gather of equal-sized band buffers reproduces the picture in place (the last band is padded
below the picture), so the next reference is one contiguous device buffer on every rank.
"""


def band_rows(mbh, world, rank):
    """(row0, row1, B): macroblock rows [row0, row1) of this rank and the common band height B."""
    band = -(-mbh // world)
    row0, row1 = min(rank * band, mbh), min((rank + 1) * band, mbh)
    return row0, row1, band


def gather_buffers(torch, world, band, width, chroma_rows_per_mb, chroma_width, device):
    """Full-picture gather buffers (Y, U, V) sized world*band macroblock rows."""
    gy = torch.zeros((world * band * 16, width), dtype=torch.uint8, device=device)
    gu = torch.zeros((world * band * chroma_rows_per_mb, chroma_width), dtype=torch.uint8, device=device)
    return gy, gu, torch.zeros_like(gu)


def band_views(bufs, rank, band, chroma_rows_per_mb):
    """This rank's send slices inside the gather buffers (views, no copy)."""
    gy, gu, gv = bufs
    return (gy[rank * band * 16:(rank + 1) * band * 16],
            gu[rank * band * chroma_rows_per_mb:(rank + 1) * band * chroma_rows_per_mb],
            gv[rank * band * chroma_rows_per_mb:(rank + 1) * band * chroma_rows_per_mb])


def send_buffers(torch, band, width, chroma_rows_per_mb, chroma_width, device):
    """This rank's band as its own contiguous tensors (Y, U, V): the all-gather's send side (not aliased with the gather buffer)."""
    sy = torch.zeros((band * 16, width), dtype=torch.uint8, device=device)
    su = torch.zeros((band * chroma_rows_per_mb, chroma_width), dtype=torch.uint8, device=device)
    return sy, su, torch.zeros_like(su)


def all_gather_recon(dist, bufs, sends):
    """One all-gather per plane: after it every rank holds the whole reconstructed picture (+ padding rows)."""
    for g, s in zip(bufs, sends):
        dist.all_gather_into_tensor(g, s)
