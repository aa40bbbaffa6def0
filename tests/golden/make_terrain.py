#!/usr/bin/env python3
"""Pins the reference's terrain cloud and body lattice as a committed fixture.

Run HERE (the build container; /root/reference does not exist on the GPU box):

    python tests/golden/make_terrain.py        ->  tests/golden/terrain_ground.npz

Imports the reference's own generator `maps.py` (numpy only; maps.py:190-297 builds `ground`: a 256 x 256 raster over
x in [-2000, 2000], y in [-6000, 2000] mm, 50 random rocks, a crater, a cliff, two boulders and two octaves of
Perlin fractal noise, np.random.seed(42)) and stores its OUTPUT -- data, not source:

  ground      (65536, 3) float32   maps.ground: the foothold cloud before.py:19-22 writes to numpy_input_t{x,y,z}.bin
  lattice_x/y/z                     the axes of before.py:24-37's body lattice (voxel 50 mm over the cloud's bounding
                                    box, z up to max + 350): the full lattice is their meshgrid (before.py:47-56)
  bodies      (~1e5, 3) float32     every lattice node within [0, 350] mm above the nearest ground sample (the only
                                    ones that can stand), in lattice order: BASELINE config 3's 1e5 body poses

before.py itself is NOT imported (it writes files into the working directory at import time); its three
np.arange calls are restated below.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: this script runs in the build container only")
    sys.path.insert(0, REF)
    import maps  # the reference's generator (numpy only)

    ground = np.ascontiguousarray(maps.ground, dtype=np.float32)
    assert ground.shape == (65536, 3)
    # before.py:24-37
    side_margin, voxel = 0, 50
    lx = np.arange(ground[:, 0].min() - side_margin, (ground[:, 0].max() + side_margin) / 1, voxel)
    ly = np.arange(ground[:, 1].min() - side_margin, (ground[:, 1].max() + side_margin) / 1, voxel) + 0
    lz = np.arange(ground[:, 2].min(), (ground[:, 2].max() + 350) / 1, voxel)
    # lattice nodes standing 0..350 mm above the nearest raster sample (the raster is regular in x, y)
    side = 256
    gz = ground[:, 2].reshape(side, side)  # meshgrid(x, y): row = y index, column = x index
    x0, x1, y0, y1 = -2000.0, 2000.0, -6000.0, 2000.0
    ix = np.clip(np.rint((lx - x0) / (x1 - x0) * (side - 1)).astype(int), 0, side - 1)
    iy = np.clip(np.rint((ly - y0) / (y1 - y0) * (side - 1)).astype(int), 0, side - 1)
    base = gz[np.ix_(iy, ix)]  # (ny, nx)
    X, Y, Z = np.meshgrid(lx, ly, lz)  # before.py:47: (ny, nx, nz)
    near = (Z >= base[:, :, None]) & (Z <= base[:, :, None] + 350.0)
    cand = np.stack([X[near], Y[near], Z[near]], -1).astype(np.float32)
    bodies = np.ascontiguousarray(cand)  # all of them: ~1e5
    out = os.path.join(HERE, "terrain_ground.npz")
    np.savez_compressed(out, ground=ground, lattice_x=lx.astype(np.float32), lattice_y=ly.astype(np.float32),
                        lattice_z=lz.astype(np.float32), bodies=bodies)
    print(f"{out}: ground {ground.shape}, lattice {len(lx)} x {len(ly)} x {len(lz)}, {len(cand)} nodes near the ground, "
          f"{len(bodies)} kept; ground[1000, 2] = {ground[1000, 2]:.4f}; {os.path.getsize(out) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
