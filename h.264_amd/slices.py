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


# ---- the one-chunk exchange bench.py uses: each rank's band travels as ONE buffer [Y rows | U rows | V rows] (jmhip_recon_pack_band),
# one all_gather_into_tensor moves all chunks, jmhip_ref_unpack_bands scatters them into the reference planes. Host mirrors of that
# layout (numpy, no device): the wire format's definition for tests and for hosts that stage the exchange through CPU memory.

def chunk_bytes(width, chroma_width, chroma_rows_per_mb, band):
    """jmhip_band_chunk_bytes: band*16 luma rows + 2 x band*chroma_rows_per_mb chroma rows."""
    return band * 16 * width + 2 * band * chroma_rows_per_mb * chroma_width


def pack_band_host(np, Y, U, V, rank, band, chroma_rows_per_mb):
    """This rank's rows of the picture as one chunk (rows past the picture's end stay zero: padding of the last band)."""
    H, W = Y.shape
    Hc, Wc = U.shape
    out = np.zeros(chunk_bytes(W, Wc, chroma_rows_per_mb, band), np.uint8)
    y0, c0 = rank * band * 16, rank * band * chroma_rows_per_mb
    ny, nc = max(0, min(band * 16, H - y0)), max(0, min(band * chroma_rows_per_mb, Hc - c0))
    ysz, csz = band * 16 * W, band * chroma_rows_per_mb * Wc
    out[:ny * W] = Y[y0:y0 + ny].reshape(-1)
    out[ysz:ysz + nc * Wc] = U[c0:c0 + nc].reshape(-1)
    out[ysz + csz:ysz + csz + nc * Wc] = V[c0:c0 + nc].reshape(-1)
    return out


def unpack_bands_host(np, chunks, world, band, H, W, Hc, Wc, chroma_rows_per_mb):
    """`world` chunks (the gathered buffer) -> the whole picture (Y, U, V)."""
    n = chunk_bytes(W, Wc, chroma_rows_per_mb, band)
    Y, U, V = np.zeros((H, W), np.uint8), np.zeros((Hc, Wc), np.uint8), np.zeros((Hc, Wc), np.uint8)
    ysz, csz = band * 16 * W, band * chroma_rows_per_mb * Wc
    for r in range(world):
        c = chunks[r * n:(r + 1) * n]
        y0, c0 = r * band * 16, r * band * chroma_rows_per_mb
        ny, nc = max(0, min(band * 16, H - y0)), max(0, min(band * chroma_rows_per_mb, Hc - c0))
        Y[y0:y0 + ny] = c[:ny * W].reshape(ny, W)
        U[c0:c0 + nc] = c[ysz:ysz + nc * Wc].reshape(nc, Wc)
        V[c0:c0 + nc] = c[ysz + csz:ysz + csz + nc * Wc].reshape(nc, Wc)
    return Y, U, V
