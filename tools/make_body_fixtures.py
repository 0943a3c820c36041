#!/usr/bin/env python
"""Voxelise the reference's Dragon_50k.obj into small point-set fixtures (container only: needs
/root/reference).  The GPU box has no /root/reference, so the full-size parity scenes load these:
  sph_taichi_amd/data/bodies/dragon_bath_body.npy     dragon as in data/scenes/dragon_bath.json (scale 1, rot 0), untranslated
  sph_taichi_amd/data/bodies/armadillo_standin.npy    dragon scaled 0.65, rotated 180 deg about y: stand-in for the missing
                                        armadillo_small.obj of armadillo_bath_dynamic.json (~5.4 k particles)
Points are stored WITHOUT the scene translation; the loader adds it (translations that are multiples of
d = 0.02 keep the voxel lattice of the original recipe; dragon_bath's (3.5, 0.05, 1.0) is applied before
voxelising, so its fixture stores points - translation)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sph_taichi_amd import voxelizer  # noqa: E402

REF = "/root/reference"
OBJ = os.path.join(REF, "data/models/Dragon_50k.obj")
out = os.path.join(ROOT, "sph_taichi_amd", "data", "bodies")

body = {"geometryFile": OBJ, "scale": [1, 1, 1], "translation": [3.5, 0.05, 1.0], "rotationAxis": [0, 1, 0],
        "rotationAngle": 0}
pts, _ = voxelizer.load_rigid_body(body, 0.02)
np.save(os.path.join(out, "dragon_bath_body.npy"), (pts - np.array(body["translation"])).astype(np.float32))
print("dragon_bath_body", pts.shape)

body = {"geometryFile": OBJ, "scale": [0.65, 0.65, 0.65], "translation": [0.0, 0.0, 0.0], "rotationAxis": [0, 1, 0],
        "rotationAngle": 180}
pts, _ = voxelizer.load_rigid_body(body, 0.02)
pts = pts - np.round(pts.mean(axis=0) / 0.02) * 0.02          # centre near the origin, on the lattice
np.save(os.path.join(out, "armadillo_standin.npy"), pts.astype(np.float32))
print("armadillo_standin", pts.shape, pts.min(0), pts.max(0))
