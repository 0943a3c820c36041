"""trimesh-free rigid-body ingestion: OBJ/STL loader, the scale /
rotate / translate sequence of particle_system.py:421-431, and a restatement of
`mesh.voxelized(pitch).fill().points` (particle_system.py:441-444; algorithm of
trimesh.voxel.creation.voxelize_subdivide + binary hole filling).

Voxel-set equality with trimesh itself cannot be checked here (trimesh is absent from
this image).  What IS checked (tests/test_voxelizer_crosscheck.py, against the test tree's
second, differently built implementation of the same published recipe plus a geometric
inside / outside classifier): identical voxel SETS on the reference's Dragon_50k.obj as every
scene file places it, on bunny_sparse.obj and on rotated / stretched cubes.  Scenes that need
bit-level ingestion parity with a trimesh installation should ship the point set
(`"voxelizedPointsFile": x.npy` in the body).
"""
from __future__ import annotations

import os
import struct

import numpy as np


class TriMesh:
    """The few members of trimesh.Trimesh the reference touches."""

    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def copy(self):
        return TriMesh(self.vertices.copy(), self.faces.copy())

    def apply_scale(self, scale):
        self.vertices = self.vertices * np.asarray(scale, dtype=np.float64)

    def apply_transform(self, M):
        v = np.c_[self.vertices, np.ones(len(self.vertices))] @ np.asarray(M).T
        self.vertices = v[:, :3]

    def export(self, file_type="obj"):
        assert file_type == "obj"
        lines = [f"v {x:.8f} {y:.8f} {z:.8f}" for x, y, z in self.vertices]
        lines += [f"f {a + 1} {b + 1} {c + 1}" for a, b, c in self.faces]
        return "\n".join(lines) + "\n"


def load_mesh(path: str) -> TriMesh:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        vs, fs = [], []
        with open(path, "r") as fh:
            for line in fh:
                if line.startswith("v "):
                    vs.append([float(t) for t in line.split()[1:4]])
                elif line.startswith("f "):
                    idx = [int(t.split("/")[0]) for t in line.split()[1:]]
                    idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                    for k in range(1, len(idx) - 1):  # fan-triangulate polygons
                        fs.append([idx[0], idx[k], idx[k + 1]])
        return TriMesh(np.array(vs), np.array(fs))
    if ext == ".stl":
        with open(path, "rb") as fh:
            data = fh.read()
        n = struct.unpack_from("<I", data, 80)[0]
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                            count=n, offset=84)
        v = rec["v"].reshape(-1, 3).astype(np.float64)
        uniq, inv = np.unique(v, axis=0, return_inverse=True)
        return TriMesh(uniq, inv.reshape(-1, 3))
    raise ValueError(f"unsupported geometry file {path}")


def rotation_matrix(angle, direction, point):
    """4x4 rotation about the axis `direction` through `point` (Rodrigues)."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    s, c = np.sin(angle), np.cos(angle)
    R = np.diag([c, c, c]) + np.outer(d, d) * (1.0 - c)
    d = d * s
    R += np.array([[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]])
    M = np.identity(4)
    M[:3, :3] = R
    p = np.asarray(point, dtype=np.float64)
    M[:3, 3] = p - R @ p
    return M


def subdivide_to_size(vertices, faces, max_edge, max_iter=10):
    """All vertices of a subdivision of the mesh whose edges are <= max_edge."""
    tri = vertices[faces]  # (F,3,3)
    out = [vertices]
    for _ in range(max_iter + 1):
        if len(tri) == 0:
            break
        e = np.stack([tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 1], tri[:, 0] - tri[:, 2]], axis=1)
        too_long = (np.linalg.norm(e, axis=2) > max_edge).any(axis=1)
        tri = tri[too_long]
        if len(tri) == 0:
            break
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = (a + b) * 0.5, (b + c) * 0.5, (c + a) * 0.5
        out.append(np.concatenate([ab, bc, ca]))
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                              np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
    return np.concatenate(out)


def voxelize_filled_points(mesh: TriMesh, pitch: float) -> np.ndarray:
    from scipy import ndimage
    v = subdivide_to_size(mesh.vertices, mesh.faces, max_edge=pitch / 2.0)
    hit = np.round(v / pitch).astype(np.int64)
    occ = np.unique(hit, axis=0)
    origin = occ.min(axis=0)
    shape = occ.max(axis=0) - origin + 1
    dense = np.zeros(shape, dtype=bool)
    dense[tuple((occ - origin).T)] = True
    dense = ndimage.binary_fill_holes(dense)
    idx = np.argwhere(dense)
    return (idx + origin) * pitch


def load_rigid_body(rigid_body: dict, particle_diameter: float, base_dir: str = "."):
    """particle_system.py:421-447.  Returns (voxel points f64 [n,3], mesh backup)."""
    d = particle_diameter
    pts_file = rigid_body.get("voxelizedPointsFile")
    if pts_file and not os.path.isabs(pts_file) and not os.path.exists(pts_file):
        pts_file = os.path.join(base_dir, pts_file)
    path = rigid_body.get("geometryFile")
    if path and not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(base_dir, path)
    offset = np.array(rigid_body["translation"], dtype=np.float64)
    mesh = backup = None
    if path and os.path.exists(path):
        mesh = load_mesh(path)
        mesh.apply_scale(rigid_body["scale"])
        angle = rigid_body["rotationAngle"] / 360 * 2 * 3.1415926   # particle_system.py:427
        direction = rigid_body["rotationAxis"]
        mesh.apply_transform(rotation_matrix(angle, direction, mesh.vertices.mean(axis=0)))
        mesh.vertices = mesh.vertices + offset
        backup = mesh.copy()
    elif not pts_file:
        raise FileNotFoundError(f"rigid body geometry {path!r} not found and no voxelizedPointsFile given")
    if pts_file:
        # pre-voxelised point set (scale/rotation already applied, translation not): exchanges the
        # trimesh-dependent voxel set as a fixture
        pts = np.load(pts_file).astype(np.float64) + offset
    else:
        pts = voxelize_filled_points(mesh, d)
    if backup is None:
        backup = TriMesh(pts.copy(), np.zeros((0, 3), dtype=np.int64))   # no mesh: export the particle cloud
    return pts, backup
