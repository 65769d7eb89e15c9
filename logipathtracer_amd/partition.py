"""Pixel ownership of the tile-sharded image, restated in numpy.

Mirrors the device-side maps (csrc/lpt_device.h: slot_to_pixel / pixel_to_slot): the image is cut into
tile_w x tile_h tiles, tile g belongs to rank g % n_ranks, and a rank's pixels live in a compact buffer
ordered by local tile, then 8x8 pixel block (row-major inside the tile), then row-major inside the block.
Used by host-side code that packs or unpacks gathered tile buffers, and by the tests.
"""
import numpy as np


class Partition:
    def __init__(self, width, height, tile_w=None, tile_h=None, n_ranks=1):
        self.W, self.H = int(width), int(height)
        self.tile_w = int(tile_w) if tile_w else (self.W + 7) & ~7
        self.tile_h = int(tile_h) if tile_h else (self.H + 7) & ~7
        if self.tile_w % 8 or self.tile_h % 8:
            raise ValueError("tile size must be a multiple of 8")
        self.n_ranks = int(n_ranks)
        self.tiles_x = -(-self.W // self.tile_w)
        self.tiles_y = -(-self.H // self.tile_h)
        self.n_local_tiles = -(-(self.tiles_x * self.tiles_y) // self.n_ranks)
        self.n_slots = self.n_local_tiles * self.tile_w * self.tile_h

    def pixel_to_slot(self, x, y):
        """(owner rank, slot) of pixel(s) (x, y)."""
        x = np.asarray(x, dtype=np.int64)
        y = np.asarray(y, dtype=np.int64)
        tx, ty = x // self.tile_w, y // self.tile_h
        g = ty * self.tiles_x + tx
        owner, k = g % self.n_ranks, g // self.n_ranks
        lx, ly = x - tx * self.tile_w, y - ty * self.tile_h
        blk = (ly >> 3) * (self.tile_w >> 3) + (lx >> 3)
        slot = k * (self.tile_w * self.tile_h) + blk * 64 + (ly & 7) * 8 + (lx & 7)
        return owner, slot

    def owned_tiles(self, rank):
        """Pixel rectangles (x0, y0, w, h), clipped to the image, of the tiles `rank` owns."""
        out = []
        for g in range(rank, self.tiles_x * self.tiles_y, self.n_ranks):
            tx, ty = g % self.tiles_x, g // self.tiles_x
            x0, y0 = tx * self.tile_w, ty * self.tile_h
            out.append((x0, y0, min(self.tile_w, self.W - x0), min(self.tile_h, self.H - y0)))
        return out

    def pack(self, image, rank):
        """Scan-line image (H, W, C) -> this rank's compact buffer (n_slots, C); unowned slots stay 0."""
        image = np.asarray(image)
        ys, xs = np.mgrid[0:self.H, 0:self.W]
        owner, slot = self.pixel_to_slot(xs, ys)
        buf = np.zeros((self.n_slots,) + image.shape[2:], dtype=image.dtype)
        m = owner == rank
        buf[slot[m]] = image[m]
        return buf

    def assemble(self, gathered):
        """Concatenated compact buffers of all ranks (n_ranks * n_slots, C) -> scan-line image (H, W, C)."""
        gathered = np.asarray(gathered).reshape(self.n_ranks * self.n_slots, -1)
        ys, xs = np.mgrid[0:self.H, 0:self.W]
        owner, slot = self.pixel_to_slot(xs, ys)
        return gathered[owner * self.n_slots + slot]
