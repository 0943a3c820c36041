"""TEST INFRASTRUCTURE: the independent check of the mesh -> particle path (voxel_check.c has the four kernels and the
references; this file is the host side -- its own OBJ / STL reader, its own restatement of the transform order of
/root/reference/particle_system.py:424-431, the comparison and its classification).  Nothing here imports
sph_taichi_amd/voxelizer.py except `compare()`, which takes its OUTPUT as an argument.

  reference(body, pitch)      -> voxel points by the second implementation (sampled shell + flood fill), f64 [n, 3]
  geometric(verts, faces, ...)-> the geometric picture of the same body: cube-touches-surface / centre-inside masks
  compare(points, body, pitch)-> report dict: are the voxel SETS identical, and what every voxel is geometrically
  mesh_audit(path)            -> what trimesh.load(process=True) / repair.fill_holes could change on this file
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libvoxel_check.so")
    src = os.path.join(_HERE, "voxel_check.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libvoxel_check.so"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip, lp, up = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_ubyte)
        L.vc_sample_shell.argtypes = [dp, ip, C.c_int, lp, ip, C.c_double, C.c_double, C.c_int, up]
        L.vc_sample_shell.restype = C.c_int
        L.vc_fill.argtypes = [up, ip, up]
        L.vc_fill.restype = None
        L.vc_box_shell.argtypes = [dp, ip, C.c_int, lp, ip, C.c_double, C.c_double, up]
        L.vc_box_shell.restype = None
        L.vc_parity.argtypes = [dp, ip, C.c_int, lp, ip, C.c_double, C.c_int, dp, up]
        L.vc_parity.restype = None
        _LIB = L
    return _LIB


# ---- reading and placing the mesh (particle_system.py:421-431) ---------------------------------------------------------
def read_mesh(path):
    """(vertices f64 [n, 3], triangles i32 [m, 3]).  OBJ: `v` and `f` records only, polygons as fans, negative indices
    relative; binary STL: vertices merged by exact position (what trimesh.load does with an STL's triangle soup)."""
    if path.lower().endswith(".stl"):
        raw = open(path, "rb").read()
        (n,) = struct.unpack_from("<I", raw, 80)
        tri = np.empty((n, 3, 3), dtype=np.float64)
        for t in range(n):
            vals = struct.unpack_from("<12f", raw, 84 + 50 * t)
            tri[t] = np.array(vals[3:12]).reshape(3, 3)
        flat = tri.reshape(-1, 3)
        keys = {}
        idx = np.empty(len(flat), dtype=np.int32)
        verts = []
        for q, p in enumerate(map(tuple, flat)):
            if p not in keys:
                keys[p] = len(verts)
                verts.append(p)
            idx[q] = keys[p]
        return np.array(verts, dtype=np.float64), idx.reshape(-1, 3)
    verts, faces = [], []
    for line in open(path, "r"):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
        elif tok[0] == "f":
            ids = []
            for t in tok[1:]:
                i = int(t.split("/")[0])
                ids.append(i - 1 if i > 0 else len(verts) + i)
            for k in range(1, len(ids) - 1):
                faces.append((ids[0], ids[k], ids[k + 1]))
    return np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32)


def place(verts, body):
    """mesh.apply_scale(scale); rotate by rotationAngle / 360 * 2 * 3.1415926 about rotationAxis through the mean of the
    (scaled) vertices; vertices += translation  (particle_system.py:423-431).  The rotation is written with the unit
    quaternion's sandwich product (voxelizer.py builds a 4 x 4 Rodrigues matrix)."""
    v = verts * np.asarray(body.get("scale", [1, 1, 1]), dtype=np.float64)
    angle = body.get("rotationAngle", 0) / 360 * 2 * 3.1415926        # the reference's own pi (:427)
    axis = np.asarray(body.get("rotationAxis", [0, 1, 0]), dtype=np.float64)
    axis = axis / np.sqrt((axis * axis).sum())
    centre = v.mean(axis=0)
    qw, qv = np.cos(angle / 2), np.sin(angle / 2) * axis
    r = v - centre
    t = 2.0 * np.cross(qv, r)
    r = r + qw * t + np.cross(qv, t)
    return r + centre + np.asarray(body.get("translation", [0, 0, 0]), dtype=np.float64)


def load_body(body, base_dir="."):
    path = body["geometryFile"]
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(base_dir, path)
    verts, faces = read_mesh(path)
    return np.ascontiguousarray(place(verts, body)), np.ascontiguousarray(faces, dtype=np.int32)


# ---- the grids ----------------------------------------------------------------------------------------------------------
def _grid(verts, pitch, margin=2):
    lo = np.floor(verts.min(axis=0) / pitch).astype(np.int64) - margin
    hi = np.ceil(verts.max(axis=0) / pitch).astype(np.int64) + margin
    return lo, (hi - lo + 1).astype(np.int32)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _args(verts, faces, lo, dims):
    o = (C.c_long * 3)(*[int(x) for x in lo])
    n = (C.c_int * 3)(*[int(x) for x in dims])
    return _ptr(verts, C.c_double), _ptr(faces, C.c_int), int(len(faces)), o, n


def sampled_filled(verts, faces, pitch, lo, dims):
    """trimesh's recipe, second implementation: boolean grids (sampled shell, shell + filled holes)."""
    L = lib()
    total = int(np.prod(dims))
    shell = np.zeros(total, dtype=np.uint8)
    v, f, nf, o, n = _args(verts, faces, lo, dims)
    if L.vc_sample_shell(v, f, nf, o, n, float(pitch), 2.0, 10, _ptr(shell, C.c_ubyte)):
        raise ValueError("max_iter exceeded!")      # trimesh.remesh.subdivide_to_size raises the same
    filled = np.zeros(total, dtype=np.uint8)
    L.vc_fill(_ptr(shell, C.c_ubyte), n, _ptr(filled, C.c_ubyte))
    return shell.reshape(dims).astype(bool), filled.reshape(dims).astype(bool)


def geometric(verts, faces, pitch, lo, dims):
    """(cube touches the surface, centre inside by the majority of three axis-parallel ray parities, the three parities)."""
    L = lib()
    total = int(np.prod(dims))
    v, f, nf, o, n = _args(verts, faces, lo, dims)
    touch = np.zeros(total, dtype=np.uint8)
    # (half width + 1e-9: a subdivision vertex that numpy.round files under a voxel lies in its CLOSED cube up to the rounding of
    # one f64 division -- the dragon has one such vertex on a cube face)
    L.vc_box_shell(v, f, nf, o, n, float(pitch), float(pitch) / 2 * (1.0 + 1e-9), _ptr(touch, C.c_ubyte))
    eps = np.array([0.0123456789, 0.0271828183, 0.0314159265]) * 1e-3 * pitch     # generic, far below pitch / 2
    par = []
    for axis in range(3):
        p = np.zeros(total, dtype=np.uint8)
        L.vc_parity(v, f, nf, o, n, float(pitch), axis, _ptr(eps, C.c_double), _ptr(p, C.c_ubyte))
        par.append(p.reshape(dims).astype(bool))
    votes = par[0].astype(np.int8) + par[1] + par[2]
    return touch.reshape(dims).astype(bool), votes >= 2, par


def reference(body, pitch, base_dir="."):
    verts, faces = load_body(body, base_dir)
    lo, dims = _grid(verts, pitch)
    _, filled = sampled_filled(verts, faces, pitch, lo, dims)
    return (np.argwhere(filled) + lo) * pitch


def compare(points, body, pitch, base_dir="."):
    """`points` = what the path under test produced for `body` (f64 [n, 3], multiples of pitch).  Returns the report."""
    verts, faces = load_body(body, base_dir)
    lo, dims = _grid(verts, pitch)
    shell, filled = sampled_filled(verts, faces, pitch, lo, dims)
    touch, inside, par = geometric(verts, faces, pitch, lo, dims)
    idx = np.rint(np.asarray(points, dtype=np.float64) / pitch).astype(np.int64) - lo
    got = np.zeros(dims, dtype=bool)
    ok = ((idx >= 0) & (idx < dims)).all(axis=1)
    got[tuple(idx[ok].T)] = True
    off_lattice = float(np.abs(np.asarray(points) - (idx + lo) * pitch).max()) if len(points) else 0.0
    settled = ~touch                                  # cubes the surface does not touch: one side of it as a whole
    disagree = settled & ~((par[0] == par[1]) & (par[1] == par[2]))
    return {
        "voxels_under_test": int(got.sum()), "voxels_outside_the_grid": int((~ok).sum()), "off_lattice_max": off_lattice,
        "voxels_second_implementation": int(filled.sum()), "sampled_shell": int(shell.sum()),
        "identical_sets": bool((got == filled).all() and ok.all()),
        "only_under_test": int((got & ~filled).sum()), "only_second_implementation": int((filled & ~got).sum()),
        # the geometric picture of the set under test
        "shell_voxels_whose_cube_the_surface_does_not_touch": int((shell & ~touch).sum()),          # must be 0
        "interior_centres_missing": int((inside & settled & ~got).sum()),                              # must be 0
        "filled_interior": int((got & settled & inside).sum()),
        "filled_on_the_surface": int((got & touch).sum()),
        "enclosed_exterior_pockets": int((got & settled & ~inside).sum()),   # outside the mesh, cut off from the outside by shell voxels
        "touched_cubes_not_in_the_set": int((touch & ~got).sum()),           # clipped by the surface, no subdivision vertex inside, centre not enclosed
        "touched_cubes_not_in_the_set_with_centre_inside": int((touch & ~got & inside).sum()),
        "ray_axes_disagree_off_the_surface": int(disagree.sum()),            # > 0 only for a mesh that is not closed
        "grid": [int(x) for x in dims],
    }


def mesh_audit(path):
    """What trimesh.load(path) (process=True: vertices merged by position, unreferenced ones dropped) and
    repair.fill_holes (boundary loops of three or four edges get faces) could change on this file, i.e. what the
    vertex mean (the rotation centre, particle_system.py:428) and the sampled shell may depend on beyond the file's own
    v / f records."""
    verts, faces = read_mesh(path)
    ref = np.zeros(len(verts), dtype=bool)
    ref[faces.ravel()] = True
    uniq = np.unique(verts, axis=0)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    es = np.sort(e, axis=1)
    ue, cnt = np.unique(es, axis=0, return_counts=True)
    boundary = ue[cnt == 1]
    blen = np.sqrt(((verts[boundary[:, 0]] - verts[boundary[:, 1]]) ** 2).sum(axis=1)) if len(boundary) else np.zeros(0)
    return {"vertices": int(len(verts)), "faces": int(len(faces)), "unreferenced_vertices": int((~ref).sum()),
            "duplicate_positions": int(len(verts) - len(uniq)), "boundary_edges": int(len(boundary)),
            "edges_shared_by_more_than_two_faces": int((cnt > 2).sum()),
            "longest_boundary_edge": float(blen.max()) if len(blen) else 0.0}
