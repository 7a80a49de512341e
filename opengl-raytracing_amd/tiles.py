"""Host mirror of the framebuffer tile layout (csrc/rt_frame.hpp) used for the multi-GPU gather.

The frame is cut into 16x16 tiles, globalTile = tileY*tilesX + (tileX + rowShift(tileY)) % tilesX with rowShift = 0 for one
rank and 11*tileY mod tilesX for several (so that a rank owns scattered tiles, not fixed columns, when tilesX is a multiple of
the world size); rank r of n owns tiles with
globalTile % n == r, stored densely (localTile = globalTile // n), 256 pixels per tile in four 8x8
quadrants (slot = q*64 + ly*8 + lx).  Every rank contributes an equally sized block (padded to
ceil(nTiles/n) tiles) so one gather moves the frame; `assemble` is the numpy statement of the
device un-tiling kernel (k_assemble in csrc/rt_api.hip)."""
from __future__ import annotations

import numpy as np

TILE = 16
TILE_PIXELS = 256
ROW_SHIFT = 11   # csrc/rt_frame.hpp kTileRowShift


def geometry(w, h, world=1):
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    n = tx * ty
    return {"W": w, "H": h, "tilesX": tx, "tilesY": ty, "nTiles": n, "world": world, "maxLocalTiles": (n + world - 1) // world}


def slot_map(w, h, world=1):
    """-> (owner[H,W], slot[H,W]): which rank owns each pixel and its slot in that rank's local buffer."""
    g = geometry(w, h, world)
    y, x = np.mgrid[0:h, 0:w]
    ty, tx = y // TILE, x // TILE
    shift = (ty * ROW_SHIFT) % g["tilesX"] if world > 1 else 0
    t = ty * g["tilesX"] + (tx + shift) % g["tilesX"]
    lx, ly = x % TILE, y % TILE
    q = (lx // 8) | ((ly // 8) << 1)
    slot = (t // world) * TILE_PIXELS + q * 64 + (ly % 8) * 8 + (lx % 8)
    return (t % world).astype(np.int32), slot.astype(np.int64)


def owner_mask(w, h, rank, world):
    owner, _ = slot_map(w, h, world)
    return (owner == rank).astype(np.uint8)


def pack_local(image, rank, world):
    """Row-major HxWxC image -> this rank's tile-major block [maxLocalTiles*256, C] (zeros where not owned / padding)."""
    h, w, c = image.shape
    g = geometry(w, h, world)
    owner, slot = slot_map(w, h, world)
    out = np.zeros((g["maxLocalTiles"] * TILE_PIXELS, c), image.dtype)
    m = owner == rank
    out[slot[m]] = image[m]
    return out


def assemble(blocks, w, h):
    """blocks[rank] = that rank's local block -> row-major HxWxC frame."""
    world = len(blocks)
    owner, slot = slot_map(w, h, world)
    stacked = np.stack(blocks, 0)
    return stacked[owner, slot]
