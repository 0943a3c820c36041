"""Stand-in for `trimesh` so /root/reference/particle_system.py (`import trimesh as tm`)
can be imported AND its load_rigid_body (particle_system.py:421-447) executed under
the taichi shim -- TEST INFRASTRUCTURE.

trimesh is not installable in this image.  The few members the reference touches
(load, apply_scale, transformations.rotation_matrix, apply_transform, copy,
repair.fill_holes, voxelized(pitch).fill().points) are provided by the package's own
trimesh-free ingestion (sph_taichi_amd/voxelizer.py).  Consequence for
the golden vectors made with it: the reference's scale / rotate / translate SEQUENCE
and everything downstream of the voxel set (rest centre of mass, compute_com,
solve_constraints, two-way coupling) are the reference's own code; the voxel set
itself is this repo's restatement of trimesh's voxeliser and is NOT pinned by trimesh.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from sph_taichi_amd import voxelizer as _v  # noqa: E402


class _Voxels:
    def __init__(self, mesh, pitch):
        self._mesh, self._pitch = mesh, pitch
        self.points = None

    def fill(self):
        out = _Voxels(self._mesh, self._pitch)
        out.points = _v.voxelize_filled_points(self._mesh, self._pitch)
        return out


class Trimesh(_v.TriMesh):
    def copy(self):
        return Trimesh(self.vertices.copy(), self.faces.copy())

    def voxelized(self, pitch):
        return _Voxels(self, pitch)


def load(path):
    m = _v.load_mesh(path)
    return Trimesh(m.vertices, m.faces)


class transformations:  # noqa: N801 (module-like namespace)
    rotation_matrix = staticmethod(_v.rotation_matrix)


class repair:  # noqa: N801
    @staticmethod
    def fill_holes(mesh):
        return True
