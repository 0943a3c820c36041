"""Host-side (NumPy) scene construction: JSON scene -> initial per-particle
arrays.  Pure host logic with no device dependency, so it is unit-testable on
CPU.  Follows ParticleSystem.__init__ of the reference (particle_system.py:
12-88 geometry/counts, :148-211 object loop, :223-235 add_particle, :450-495
cube lattices).
"""
from __future__ import annotations

import os
from functools import reduce

import numpy as np

from . import voxelizer

MATERIAL_SOLID = 0   # particle_system.py:30
MATERIAL_FLUID = 1   # particle_system.py:31

ARRAY_SPECS = {      # name -> (dtype, vector width)   particle_system.py:101-113
    "object_id": (np.int32, 0), "x": (np.float32, 3), "x_0": (np.float32, 3), "v": (np.float32, 3),
    "acceleration": (np.float32, 3), "m_V": (np.float32, 0), "m": (np.float32, 0),
    "density": (np.float32, 0), "pressure": (np.float32, 0), "material": (np.int32, 0),
    "color": (np.int32, 3), "is_dynamic": (np.int32, 0),
}


class Geometry:
    """Scalars of particle_system.py:17-46."""

    def __init__(self, cfg):
        self.domain_start = np.array(cfg.get_cfg("domainStart"))
        self.domain_end = np.array(cfg.get_cfg("domainEnd"))
        self.domain_size = self.domain_end - self.domain_start
        self.dim = len(self.domain_size)
        assert self.dim > 1
        if self.dim != 3:
            raise NotImplementedError("only 3-D scenes are supported (every reference scene is 3-D)")
        self.particle_radius = cfg.get_cfg("particleRadius")
        self.particle_diameter = 2 * self.particle_radius
        self.support_radius = self.particle_radius * 4.0
        self.m_V0 = 0.8 * self.particle_diameter ** self.dim
        self.grid_size = self.support_radius
        self.grid_num = np.ceil(self.domain_size / self.grid_size).astype(int)
        self.padding = self.grid_size


def compute_cube_particle_num(start, end, diameter, dim=3):
    """particle_system.py:450-456 (np.arange lengths)."""
    return reduce(lambda a, b: a * b, [len(np.arange(start[i], end[i], diameter)) for i in range(dim)])


def cube_positions(lower_corner, cube_size, diameter, dim=3):
    """particle_system.py:469-483: ij-meshgrid of np.arange axes, cast to f32, z fastest."""
    axes = [np.arange(lower_corner[i], lower_corner[i] + cube_size[i], diameter) for i in range(dim)]
    grid = np.array(np.meshgrid(*axes, sparse=False, indexing="ij"), dtype=np.float32)
    return grid.reshape(dim, -1).transpose().copy()


class SceneBuilder:
    """Accumulates particles object by object (the reference appends to Taichi
    fields through _add_particles; here they are NumPy chunks)."""

    def __init__(self, geom: Geometry, x_filter=None, state=None):
        """`state` = {"x": f32[N, 3], "v": f32[N, 3]} by persistent id (the order the scene file creates its particles in):
        the particles START there instead of where the scene file puts them -- a restart -- and `x_filter` (the slab a rank
        owns) is applied to the restart positions; x_0 stays the scene file's position (the rest shape of a body)."""
        self.g = geom
        self.state = None
        if state is not None:
            self.state = {k: np.ascontiguousarray(state[k], dtype=np.float32).reshape(-1, 3) for k in ("x", "v")}
        self.chunks = {k: [] for k in ARRAY_SPECS}
        self.chunks["pid"] = []
        self.count = 0          # particles kept (== global_count without a filter)
        self.global_count = 0   # particles of the whole scene so far; pid = index in the whole scene
        self.x_filter = x_filter

    def add_particles(self, object_id, n, positions, velocity, density, pressure, material, is_dynamic, color):
        """particle_system.py:223-284 (add_particle semantics: x_0 = x, m_V = m_V0, m = m_V0*density)."""
        positions = np.asarray(positions, dtype=np.float32).reshape(n, 3)
        pid = self.global_count + np.arange(n, dtype=np.int64)
        first = self.global_count
        self.global_count += n
        rest = None
        if self.state is not None:
            rest = positions
            positions = self.state["x"][first:first + n]
            velocity = self.state["v"][first:first + n]
        if self.x_filter is not None:
            keep = self.x_filter(positions[:, 0])
            sel = lambda a, w=None: np.asarray(a).reshape((n, w) if w else (n,))[keep]
            positions, pid = positions[keep], pid[keep]
            rest = rest[keep] if rest is not None else None
            velocity, color = sel(velocity, 3), sel(color, 3)
            density, pressure, material, is_dynamic = sel(density), sel(pressure), sel(material), sel(is_dynamic)
            n = positions.shape[0]
        self._append(object_id, n, positions, velocity, density, pressure, material, is_dynamic, color, pid, rest)

    def _append(self, object_id, n, positions, velocity, density, pressure, material, is_dynamic, color, pid, rest=None):
        density = np.asarray(density, dtype=np.float32).reshape(n)
        c = self.chunks
        c["pid"].append(np.asarray(pid, dtype=np.int32))
        c["object_id"].append(np.full(n, object_id, dtype=np.int32))
        c["x"].append(np.ascontiguousarray(positions, dtype=np.float32))
        c["x_0"].append(positions.copy() if rest is None else np.ascontiguousarray(rest, dtype=np.float32))
        c["v"].append(np.asarray(velocity, dtype=np.float32).reshape(n, 3))
        c["acceleration"].append(np.zeros((n, 3), dtype=np.float32))
        c["m_V"].append(np.full(n, self.g.m_V0, dtype=np.float32))
        c["m"].append((np.float32(self.g.m_V0) * density).astype(np.float32))
        self.count += n
        c["density"].append(density)
        c["pressure"].append(np.asarray(pressure, dtype=np.float32).reshape(n))
        c["material"].append(np.asarray(material, dtype=np.int32).reshape(n))
        c["is_dynamic"].append(np.asarray(is_dynamic, dtype=np.int32).reshape(n))
        c["color"].append(np.asarray(color, dtype=np.int32).reshape(n, 3))

    def add_cube(self, object_id, lower_corner, cube_size, material, is_dynamic, color=(0, 0, 0), density=None,
                 pressure=None, velocity=None):
        """particle_system.py:458-495."""
        d = self.g.particle_diameter
        axes = [np.arange(lower_corner[i], lower_corner[i] + cube_size[i], d) for i in range(self.g.dim)]
        n_full = len(axes[0]) * len(axes[1]) * len(axes[2])
        if self.state is not None and self.x_filter is not None:
            # restart of a slab rank: the filter looks at where the particles ARE; only the kept ones' lattice positions
            # (x_0) are materialised, from their persistent ids
            first = self.global_count
            self.global_count += n_full
            x = self.state["x"][first:first + n_full]
            keep = np.nonzero(self.x_filter(x[:, 0]))[0]
            ny, nz = len(axes[1]), len(axes[2])
            a32 = [a.astype(np.float32) for a in axes]
            rest = np.stack([a32[0][keep // (ny * nz)], a32[1][(keep // nz) % ny], a32[2][keep % nz]], axis=1)
            n = keep.shape[0]
            self._append(object_id, n, x[keep], self.state["v"][first:first + n_full][keep],
                         np.full(n, density if density is not None else 1000.0, dtype=np.float32),
                         np.full(n, pressure if pressure is not None else 0.0, dtype=np.float32),
                         np.full(n, material, dtype=np.int32), np.full(n, is_dynamic, dtype=np.int32),
                         np.tile(np.asarray(color, dtype=np.int32), (n, 1)), first + keep, rest)
            return n
        if self.x_filter is not None and self.state is None:
            # subset at axis level: only the x-planes of this slab are ever materialised
            keep_ix = np.nonzero(self.x_filter(axes[0].astype(np.float32)))[0]
            grid = np.array(np.meshgrid(axes[0][keep_ix], axes[1], axes[2], sparse=False, indexing="ij"),
                            dtype=np.float32)
            pos = grid.reshape(3, -1).transpose().copy()
            ny, nz = len(axes[1]), len(axes[2])
            pid = (self.global_count + (keep_ix[:, None, None].astype(np.int64) * ny
                                        + np.arange(ny)[None, :, None]) * nz + np.arange(nz)[None, None, :]).reshape(-1)
            self.global_count += n_full
            n = pos.shape[0]
            vel = np.zeros_like(pos) if velocity is None else np.tile(np.asarray(velocity, dtype=np.float32), (n, 1))
            self._append(object_id, n, pos, vel,
                         np.full(n, density if density is not None else 1000.0, dtype=np.float32),
                         np.full(n, pressure if pressure is not None else 0.0, dtype=np.float32),
                         np.full(n, material, dtype=np.int32), np.full(n, is_dynamic, dtype=np.int32),
                         np.tile(np.asarray(color, dtype=np.int32), (n, 1)), pid)
            return n
        pos = cube_positions(lower_corner, cube_size, self.g.particle_diameter, self.g.dim)
        n = pos.shape[0]
        vel = np.zeros_like(pos) if velocity is None else np.tile(np.asarray(velocity, dtype=np.float32), (n, 1))
        self.add_particles(object_id, n, pos, vel,
                           np.full(n, density if density is not None else 1000.0, dtype=np.float32),
                           np.full(n, pressure if pressure is not None else 0.0, dtype=np.float32),
                           np.full(n, material, dtype=np.int32), np.full(n, is_dynamic, dtype=np.int32),
                           np.tile(np.asarray(color, dtype=np.int32), (n, 1)))
        return n

    def arrays(self):
        out = {}
        for k, (dt, vec) in ARRAY_SPECS.items():
            shape = (0, vec) if vec else (0,)
            out[k] = np.concatenate(self.chunks[k]).astype(dt) if self.chunks[k] else np.zeros(shape, dtype=dt)
        out["pid"] = np.concatenate(self.chunks["pid"]).astype(np.int32) if self.chunks["pid"] else np.zeros(0, np.int32)
        return out


class Scene:
    """Result of build_scene: geometry, object bookkeeping and initial arrays."""


def build_scene(cfg, base_dir: str | None = None, verbose: bool = False, x_filter=None, state=None) -> Scene:
    """`state`: restart positions / velocities by persistent id (SceneBuilder)."""
    g = Geometry(cfg)
    sc = Scene()
    sc.geom = g
    sc.object_collection = {}
    sc.object_id_rigid_body = set()
    base_dir = base_dir or os.getcwd()

    # ---- particle counts (particle_system.py:52-83) ----
    fluid_blocks, rigid_blocks, rigid_bodies = cfg.get_fluid_blocks(), cfg.get_rigid_blocks(), cfg.get_rigid_bodies()
    fluid_n = 0
    for fluid in fluid_blocks:
        n = compute_cube_particle_num(fluid["start"], fluid["end"], g.particle_diameter, g.dim)
        fluid["particleNum"] = n
        sc.object_collection[fluid["objectId"]] = fluid
        fluid_n += n
    rigid_n = 0
    for rigid in rigid_blocks:
        n = compute_cube_particle_num(rigid["start"], rigid["end"], g.particle_diameter, g.dim)
        rigid["particleNum"] = n
        sc.object_collection[rigid["objectId"]] = rigid
        rigid_n += n
    for body in rigid_bodies:
        pts, mesh = voxelizer.load_rigid_body(body, g.particle_diameter, base_dir)
        body["particleNum"] = pts.shape[0]
        body["voxelizedPoints"] = pts
        body["mesh"] = mesh
        body["restPosition"] = mesh.vertices
        body["restCenterOfMass"] = mesh.vertices.mean(axis=0)
        sc.object_collection[body["objectId"]] = body
        rigid_n += pts.shape[0]
        if verbose:
            print(f"rigid body {body['objectId']} num: {pts.shape[0]}")
    sc.fluid_particle_num = fluid_n
    sc.solid_particle_num = rigid_n
    sc.particle_max_num = fluid_n + rigid_n
    sc.num_rigid_bodies = len(rigid_blocks) + len(rigid_bodies)
    sc.n_objects = sc.num_rigid_bodies + len(fluid_blocks)     # len(rigid_rest_cm), particle_system.py:93

    # ---- particles (particle_system.py:148-211) ----
    if state is not None and (np.asarray(state["x"]).shape[0] != sc.particle_max_num or np.asarray(state["v"]).shape[0] != sc.particle_max_num):
        raise ValueError(f"restart state has {np.asarray(state['x']).shape[0]} rows; the scene file creates {sc.particle_max_num} particles")
    b = SceneBuilder(g, x_filter, state)
    for fluid in fluid_blocks:
        off = np.array(fluid["translation"])
        start, end = np.array(fluid["start"]) + off, np.array(fluid["end"]) + off
        b.add_cube(fluid["objectId"], start, (end - start) * np.array(fluid["scale"]), material=MATERIAL_FLUID,
                   is_dynamic=1, color=fluid["color"], density=fluid["density"], velocity=fluid["velocity"])
    for rigid in rigid_blocks:
        off = np.array(rigid["translation"])
        start, end = np.array(rigid["start"]) + off, np.array(rigid["end"]) + off
        b.add_cube(rigid["objectId"], start, (end - start) * np.array(rigid["scale"]), material=MATERIAL_SOLID,
                   is_dynamic=rigid["isDynamic"], color=rigid["color"], density=rigid["density"],
                   velocity=rigid["velocity"])
    for body in rigid_bodies:
        oid = body["objectId"]
        sc.object_id_rigid_body.add(oid)
        n = body["particleNum"]
        body["pidStart"] = b.global_count          # persistent ids of the body: pidStart + [0, n)
        dyn = body["isDynamic"]
        vel = np.array(body["velocity"], dtype=np.float32) if dyn else np.zeros(3, dtype=np.float32)
        b.add_particles(oid, n, np.array(body["voxelizedPoints"], dtype=np.float32), np.tile(vel, (n, 1)),
                        body["density"] * np.ones(n, dtype=np.float32), np.zeros(n, dtype=np.float32),
                        np.zeros(n, dtype=np.int32), int(bool(dyn)) * np.ones(n, dtype=np.int32),
                        np.tile(np.array(body["color"], dtype=np.int32), (n, 1)))
    if b.global_count != sc.particle_max_num:
        raise ValueError(f"scene is inconsistent: blocks create {b.global_count} particles but start/end predict "
                         f"{sc.particle_max_num} (the reference would overflow its fields here)")
    sc.arrays = b.arrays()
    sc.dynamic_rigid_ids = [oid for oid in sc.object_id_rigid_body if sc.object_collection[oid]["isDynamic"]]
    return sc


def kernel_constants(support_radius: float, viscosity: float, dim: int = 3):
    """Python-scope f64 folding of the reference's kernel constants
    (sph_base.py:27-35, 50-57; WCSPH.py:104, 113)."""
    k = 8 / np.pi
    k /= support_radius ** dim
    k_dw = 6.0 * (8 / np.pi) / support_radius ** dim
    return dict(k_w=k, k_dw=k_dw, visc_d_nu=2 * (dim + 2) * viscosity, visc_eps=0.01 * support_radius ** 2)


def x_layer_of(xs, grid_size, nx):
    """Cell x-layer of f32 coordinates, computed exactly like the device hash
    (particle_system.py:287-289: f32 divide, truncate; clamped to the grid)."""
    layer = (np.asarray(xs, dtype=np.float32) / np.float32(grid_size)).astype(np.int64)
    return np.clip(layer, 0, nx - 1)


def x_layer_histogram(cfg, base_dir=None, state=None):
    """Particles per global x cell layer, from the block axes (no particle arrays) and the bodies' voxel points -- or, for
    a restart, from the restart positions."""
    g = Geometry(cfg)
    nx = int(g.grid_num[0])
    if state is not None:
        return np.bincount(x_layer_of(np.asarray(state["x"], dtype=np.float32)[:, 0], g.grid_size, nx), minlength=nx).astype(np.int64)
    hist = np.zeros(nx, dtype=np.int64)
    for blk in list(cfg.get_fluid_blocks()) + list(cfg.get_rigid_blocks()):
        off = np.array(blk["translation"])
        start, end = np.array(blk["start"]) + off, np.array(blk["end"]) + off
        size = (end - start) * np.array(blk["scale"])
        axes = [np.arange(start[i], start[i] + size[i], g.particle_diameter) for i in range(3)]
        np.add.at(hist, x_layer_of(axes[0], g.grid_size, nx), len(axes[1]) * len(axes[2]))
    for body in cfg.get_rigid_bodies():
        pts = body.get("voxelizedPoints")
        if pts is None:
            pts, _ = voxelizer.load_rigid_body(body, g.particle_diameter, base_dir or os.getcwd())
        np.add.at(hist, x_layer_of(np.asarray(pts, dtype=np.float32)[:, 0], g.grid_size, nx), 1)
    return hist


def slab_cuts(hist, world, min_width=3):
    """Cell-layer cut planes X_0=0 < X_1 < ... < X_world=nx giving every rank about the same
    number of particles.  Each slab is at least `min_width` layers wide."""
    nx = len(hist)
    if world * min_width > nx:
        raise ValueError("too many ranks for this grid")
    cum = np.concatenate([[0], np.cumsum(hist)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        x = int(np.searchsorted(cum, total * r / world, side="left"))
        x = max(x, cuts[-1] + min_width)
        x = min(x, nx - (world - r) * min_width)
        cuts.append(x)
    cuts.append(nx)
    return cuts
