"""Screen-space shard index math of the multi-GPU path, host side, numpy only.

Bins are MTR_BIN x MTR_BIN pixels, numbered row-major.  Which rank owns a bin is the host's choice per frame
(include/mtr.h: MTR_OWN_*): interleaved (``b % world``), bands of bin rows, or super-tiles dealt round-robin.  Each rank
packs its bins, in its list order, into ``shard_bins * BIN*BIN`` RGBA8 pixels (the all-gather send buffer, padded to the
largest share); ``unpack_shards`` rebuilds the linear framebuffer from the gathered blocks.  csrc/mtr_api.cpp
(build_own_lists) and csrc/k_shard.hip implement exactly these functions for the device."""
from __future__ import annotations

import numpy as np

BIN = 16
INTERLEAVED, BANDS, SUPERTILES = 0, 1, 2


def grid(width: int, height: int):
    nbx, nby = (width + BIN - 1) // BIN, (height + BIN - 1) // BIN
    return nbx, nby, nbx * nby


def equal_bands(height: int, world: int):
    nby = (height + BIN - 1) // BIN
    return [r * nby // world for r in range(world + 1)]


def balanced_bands(row_weight, world: int):
    """bin rows -> world contiguous bands of about equal total weight (every band may be empty)"""
    w = np.asarray(row_weight, dtype=np.float64)
    nby = w.size
    if w.sum() <= 0:
        return [r * nby // world for r in range(world + 1)]
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bands = [0]
    for r in range(1, world):
        bands.append(max(bands[-1], int(np.searchsorted(cum, cum[-1] * r / world, side="left"))))
    bands.append(nby)
    return [min(b, nby) for b in bands]


def own_lists(width: int, height: int, world: int, own_map: int = INTERLEAVED, param: int = 0, band_rows=None):
    """per rank: its bins in pack / tile-kernel order"""
    nbx, nby, nbins = grid(width, height)
    per = [[] for _ in range(world)]
    if own_map == BANDS:
        bands = list(band_rows) if band_rows is not None else equal_bands(height, world)
        for r in range(world):
            per[r] = list(range(bands[r] * nbx, bands[r + 1] * nbx))
    elif own_map == SUPERTILES:
        S = 1 << param
        nsx, nsy = (nbx + S - 1) // S, (nby + S - 1) // S
        for st in range(nsx * nsy):
            sx, sy = st % nsx, st // nsx
            for by in range(sy * S, min(nby, (sy + 1) * S)):
                for bx in range(sx * S, min(nbx, (sx + 1) * S)):
                    per[st % world].append(by * nbx + bx)
    else:
        for b in range(nbins):
            per[b % world].append(b)
    return per


def shard_bins(width: int, height: int, world: int, own_map: int = INTERLEAVED, param: int = 0, band_rows=None) -> int:
    return max(len(p) for p in own_lists(width, height, world, own_map, param, band_rows))


def shard_bytes(width: int, height: int, world: int, own_map: int = INTERLEAVED, param: int = 0, band_rows=None) -> int:
    return shard_bins(width, height, world, own_map, param, band_rows) * BIN * BIN * 4


def owner_map(width: int, height: int, world: int, own_map: int = INTERLEAVED, param: int = 0, band_rows=None) -> np.ndarray:
    """(H, W) array: the rank that renders each pixel."""
    nbx, nby, nbins = grid(width, height)
    owner_of_bin = np.zeros(nbins, dtype=np.int64)
    for r, lst in enumerate(own_lists(width, height, world, own_map, param, band_rows)):
        owner_of_bin[lst] = r
    y, x = np.mgrid[0:height, 0:width]
    return owner_of_bin[(y // BIN) * nbx + (x // BIN)]


def pack_shard(color: np.ndarray, rank: int, world: int, own_map: int = INTERLEAVED, param: int = 0, band_rows=None) -> np.ndarray:
    """color (H, W, 4) uint8 -> (shard_bins, BIN, BIN, 4) uint8 holding this rank's bins."""
    h, w = color.shape[:2]
    nbx, _, _ = grid(w, h)
    lists = own_lists(w, h, world, own_map, param, band_rows)
    out = np.zeros((max(len(p) for p in lists), BIN, BIN, 4), dtype=np.uint8)
    for k, b in enumerate(lists[rank]):
        x0, y0 = (b % nbx) * BIN, (b // nbx) * BIN
        blk = color[y0:y0 + BIN, x0:x0 + BIN]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def unpack_shards(gathered: np.ndarray, width: int, height: int, own_map: int = INTERLEAVED, param: int = 0, band_rows=None) -> np.ndarray:
    """gathered (world, shard_bins, BIN, BIN, 4) -> (H, W, 4)."""
    world = gathered.shape[0]
    nbx, _, _ = grid(width, height)
    out = np.zeros((height, width, 4), dtype=np.uint8)
    for r, lst in enumerate(own_lists(width, height, world, own_map, param, band_rows)):
        for k, b in enumerate(lst):
            x0, y0 = (b % nbx) * BIN, (b // nbx) * BIN
            hh, ww = min(BIN, height - y0), min(BIN, width - x0)
            out[y0:y0 + hh, x0:x0 + ww] = gathered[r, k][:hh, :ww]
    return out
