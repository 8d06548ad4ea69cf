"""Screen-space shard index math of the multi-GPU path, host side, numpy only.

Bins are MTR_BIN x MTR_BIN pixels, numbered row-major; bin b belongs to rank ``b % world``.  Each
rank packs its bins bin-major into ``shard_bins * BIN*BIN`` RGBA8 pixels (the all-gather send
buffer); ``unpack_shards`` rebuilds the linear framebuffer from the gathered blocks.  The HIP
kernels in csrc/k_shard.hip implement exactly these two functions on the device."""
from __future__ import annotations

import numpy as np

BIN = 16


def grid(width: int, height: int):
    nbx, nby = (width + BIN - 1) // BIN, (height + BIN - 1) // BIN
    return nbx, nby, nbx * nby


def shard_bins(width: int, height: int, world: int) -> int:
    return (grid(width, height)[2] + world - 1) // world


def shard_bytes(width: int, height: int, world: int) -> int:
    return shard_bins(width, height, world) * BIN * BIN * 4


def owner_map(width: int, height: int, world: int) -> np.ndarray:
    """(H, W) array: the rank that renders each pixel."""
    nbx, _, _ = grid(width, height)
    y, x = np.mgrid[0:height, 0:width]
    return ((y // BIN) * nbx + (x // BIN)) % world


def pack_shard(color: np.ndarray, rank: int, world: int) -> np.ndarray:
    """color (H, W, 4) uint8 -> (shard_bins, BIN, BIN, 4) uint8 holding this rank's bins."""
    h, w = color.shape[:2]
    nbx, _, nbins = grid(w, h)
    out = np.zeros((shard_bins(w, h, world), BIN, BIN, 4), dtype=np.uint8)
    for k in range(out.shape[0]):
        b = k * world + rank
        if b >= nbins:
            break
        x0, y0 = (b % nbx) * BIN, (b // nbx) * BIN
        blk = color[y0:y0 + BIN, x0:x0 + BIN]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def unpack_shards(gathered: np.ndarray, width: int, height: int) -> np.ndarray:
    """gathered (world, shard_bins, BIN, BIN, 4) -> (H, W, 4)."""
    world = gathered.shape[0]
    nbx, _, nbins = grid(width, height)
    out = np.zeros((height, width, 4), dtype=np.uint8)
    for b in range(nbins):
        x0, y0 = (b % nbx) * BIN, (b // nbx) * BIN
        blk = gathered[b % world, b // world]
        hh, ww = min(BIN, height - y0), min(BIN, width - x0)
        out[y0:y0 + hh, x0:x0 + ww] = blk[:hh, :ww]
    return out
