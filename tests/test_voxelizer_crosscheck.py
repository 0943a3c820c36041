"""SURVEY 8 (f1), the RigidBodies half: `sph_taichi_amd/voxelizer.py` (the product's restatement of
/root/reference/particle_system.py:421-447 without trimesh) against something that is not itself (VERDICT r05 "missing" #2,
"weak" #1).  oracle/voxel_check.{c,py}: a second, differently built implementation of trimesh's published recipe (per-face
depth-first subdivision + breadth-first hole filling, its own OBJ reader and quaternion rotation) -- the voxel SETS must be
identical -- and a geometric classifier that does not know the recipe (exact triangle / cube overlap, ray parity along
three axes): every sampled voxel's cube touches the surface, every centre inside a closed mesh is filled, whatever else is
filled is an enclosed exterior pocket (counted).  Runs on generated cubes everywhere and on the reference's own models
where /root/reference exists (this container); the committed dragon fixture is pinned by hash on both sides."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import scenes
from oracle import voxel_check as vc
from sph_taichi_amd import voxelizer, workloads

REF = "/root/reference"
PITCH = 0.02
DRAGON_FIXTURE_SHA256 = "1decd5f51f525490b90eabcf7fe32854a392b9a6a4d71f97fe7ae5e206025af1"
DRAGON_FIXTURE_ROWS = 18496


def _closed_mesh_report(body, tmp_note=""):
    pts, _ = voxelizer.load_rigid_body(dict(body), PITCH)
    rep = vc.compare(pts, body, PITCH)
    assert rep["identical_sets"], (tmp_note, rep)
    assert rep["voxels_outside_the_grid"] == 0 and rep["off_lattice_max"] == 0.0
    # the geometric picture (closed meshes only)
    assert rep["ray_axes_disagree_off_the_surface"] == 0, rep
    assert rep["shell_voxels_whose_cube_the_surface_does_not_touch"] == 0, rep
    assert rep["interior_centres_missing"] == 0 and rep["touched_cubes_not_in_the_set_with_centre_inside"] == 0, rep
    assert rep["filled_interior"] + rep["filled_on_the_surface"] + rep["enclosed_exterior_pockets"] == rep["voxels_under_test"]
    return rep


def _cube_body(path, **kw):
    b = {"geometryFile": path, "translation": [0.0, 0.0, 0.0], "rotationAxis": [0, 0, 1], "rotationAngle": 0, "scale": [1, 1, 1]}
    b.update(kw)
    return b


def test_cubes_generic_and_rotated(tmp_path):
    obj = str(tmp_path / "cube.obj")
    scenes.write_cube_obj(obj, (0.0, 0.0, 0.0), 0.1)
    cases = {
        "off the lattice": _cube_body(obj, translation=[0.2037, 0.1011, 0.3023]),
        "rotated 30 deg about z": _cube_body(obj, translation=[0.28, 0.27, 0.18], rotationAngle=30),
        "oblique axis, stretched": _cube_body(obj, translation=[0.5, 0.3, 0.2], rotationAxis=[1, 2, 3], rotationAngle=131,
                                              scale=[1.7, 0.6, 1.1]),
    }
    for name, body in cases.items():
        rep = _closed_mesh_report(body, name)
        assert rep["enclosed_exterior_pockets"] == 0 and rep["filled_interior"] > 0, (name, rep)


def test_cube_on_the_lattice_ties_round_like_numpy(tmp_path):
    """The 0.1 cube at the origin: faces ON cube boundaries of the voxel lattice and subdivision vertices at x.5 pitches --
    every voxel is decided by round-half-even on an f64 quotient (0.03 / 0.02 = 1.4999999999999998).  Both implementations
    must file the ties alike: 216 voxels, identical."""
    obj = str(tmp_path / "cube.obj")
    scenes.write_cube_obj(obj, (0.0, 0.0, 0.0), 0.1)
    body = _cube_body(obj)
    pts, _ = voxelizer.load_rigid_body(dict(body), PITCH)
    rep = vc.compare(pts, body, PITCH)
    assert rep["identical_sets"] and rep["voxels_under_test"] == 216, rep
    assert rep["interior_centres_missing"] == 0


def test_transform_order_is_scale_rotate_about_the_vertex_mean_translate(tmp_path):
    """particle_system.py:423-431, with the reference's own pi = 3.1415926: the checker's quaternion placement and the
    product's 4 x 4 matrix agree to rounding, and differ visibly from any other order."""
    obj = str(tmp_path / "cube.obj")
    scenes.write_cube_obj(obj, (0.0, 0.0, 0.0), 0.1)
    body = _cube_body(obj, translation=[0.5, 0.3, 0.2], rotationAxis=[1, 2, 3], rotationAngle=131, scale=[1.7, 0.6, 1.1])
    v_check, _ = vc.load_body(body)
    _, mesh = voxelizer.load_rigid_body(dict(body), PITCH)
    assert np.abs(v_check - mesh.vertices).max() <= 1e-15 * 10
    raw, _ = vc.read_mesh(obj)
    wrong = vc.place(raw + np.array(body["translation"]), dict(body, translation=[0, 0, 0]))   # translate first: another body
    assert np.abs(wrong - v_check).max() > 1e-2
    half_turn = vc.place(raw, _cube_body(obj, rotationAngle=180, rotationAxis=[0, 1, 0]))
    assert 0.0 < np.abs(half_turn[:, 0] - (0.1 - raw[:, 0])).max() < 1e-8      # pi = 3.1415926: 5e-8 rad short of a half turn


needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "data", "models")),
                                     reason="/root/reference is not on this machine (the GPU box): the fixture hash test covers it")


def _reference_scene_bodies():
    out = []
    for path in sorted(glob.glob(os.path.join(REF, "data", "scenes", "*.json"))):
        for body in json.load(open(path)).get("RigidBodies", []):
            g = os.path.join(REF, body["geometryFile"])
            if os.path.exists(g):
                out.append((os.path.basename(path), dict(body, geometryFile=g)))
    return out


@needs_reference
def test_dragon_as_every_reference_scene_places_it():
    """Every RigidBodies entry of /root/reference/data/scenes/*.json whose mesh exists (Dragon_50k.obj: dragon_bath,
    dragon_bath_dfsph, dragon_bath_dynamic_dfsph; armadillo_small.obj is missing from the checkout), plus the stand-in the
    C3 workload uses for the armadillo (scale 0.65, half a turn about y).  Dragon_50k.obj: 25,007 vertices, none
    unreferenced, no duplicate position, no boundary edge (16 edges are shared by four faces: two sheets touching), so
    trimesh.load(process=True) keeps the vertex list -- hence the rotation centre -- and repair.fill_holes adds nothing."""
    audit = vc.mesh_audit(os.path.join(REF, "data/models/Dragon_50k.obj"))
    assert audit["unreferenced_vertices"] == 0 and audit["duplicate_positions"] == 0 and audit["boundary_edges"] == 0, audit
    bodies = _reference_scene_bodies()
    assert len(bodies) >= 3
    bodies.append(("armadillo stand-in", {"geometryFile": os.path.join(REF, "data/models/Dragon_50k.obj"), "scale": [0.65] * 3,
                                          "translation": [0.0, 0.0, 0.0], "rotationAxis": [0, 1, 0], "rotationAngle": 180}))
    seen = {}
    for name, body in bodies:
        rep = _closed_mesh_report(body, name)
        seen[name] = rep
        assert rep["enclosed_exterior_pockets"] <= 2, (name, rep)      # dragon_bath: one cube between the jaws' shell voxels
    assert seen["dragon_bath.json"]["voxels_under_test"] == DRAGON_FIXTURE_ROWS
    out = scenes.evidence_path("voxelizer_crosscheck.json")
    if out:
        json.dump(seen, open(out, "w"), indent=1)


@needs_reference
def test_open_mesh_is_voxelised_hollow_by_both_and_flagged():
    """bunny_sparse.obj has 42 boundary edges (the scanner's holes in the base, up to 0.026 long = 5 pitches at scale 4):
    the sampled shell is not closed, hole filling floods the inside, and the body is a hollow shell -- in both
    implementations alike, as trimesh's recipe has it.  The classifier reports the mesh as not closed."""
    body = {"geometryFile": os.path.join(REF, "data/models/bunny_sparse.obj"), "scale": [4, 4, 4], "translation": [1.0, 0.1, 1.0],
            "rotationAxis": [1, 1, 0], "rotationAngle": 37}
    assert vc.mesh_audit(body["geometryFile"])["boundary_edges"] == 42
    pts, _ = voxelizer.load_rigid_body(dict(body), PITCH)
    rep = vc.compare(pts, body, PITCH)
    assert rep["identical_sets"] and rep["shell_voxels_whose_cube_the_surface_does_not_touch"] == 0, rep
    assert rep["ray_axes_disagree_off_the_surface"] > 0 and rep["voxels_under_test"] < 1.01 * rep["sampled_shell"], rep


def _fixture_sha():
    a = np.load(os.path.join(workloads.BODIES, "dragon_bath_body.npy"))
    assert a.dtype == np.float32 and a.shape == (DRAGON_FIXTURE_ROWS, 3)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_committed_dragon_fixture_is_the_checked_voxel_set():
    """The body every C2 comparison stands on (sph_taichi_amd/data/bodies/dragon_bath_body.npy: 18,496 points, scene
    translation removed) cannot drift: its bytes are pinned here, and where the reference's mesh is present the SECOND
    implementation reproduces it point for point."""
    assert _fixture_sha() == DRAGON_FIXTURE_SHA256
    if not os.path.isdir(os.path.join(REF, "data", "models")):
        return
    body = [b for n, b in _reference_scene_bodies() if n == "dragon_bath.json"][0]
    pts = vc.reference(body, PITCH)
    want = np.load(os.path.join(workloads.BODIES, "dragon_bath_body.npy"))
    got = (pts - np.array(body["translation"])).astype(np.float32)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_c2_body_on_the_gpu_box_is_the_pinned_fixture():
    """The same pin where the C2 tests run (no /root/reference there): the dragon the HIP path and the oracle are compared
    on is the cross-checked set, and it reaches the device unchanged."""
    from sph_taichi_amd import ParticleSystem, SimConfig
    import copy
    assert _fixture_sha() == DRAGON_FIXTURE_SHA256
    sd = workloads.dragon_bath_scene()
    sd["FluidBlocks"] = []                                   # the body alone: 18,496 static solids
    ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
    x = ps.x.to_numpy()
    want = np.load(os.path.join(workloads.BODIES, "dragon_bath_body.npy")).astype(np.float64) + np.array([3.5, 0.05, 1.0])
    assert x.shape[0] == DRAGON_FIXTURE_ROWS and np.array_equal(x, want.astype(np.float32))
    ps.close()


@needs_reference
def test_reference_load_rigid_body_runs_on_the_independent_voxeliser_and_reproduces_the_fixtures():
    """oracle/taichi_shim/trimesh no longer delegates to the product (rounds 1-5 did: the body fixtures were compared with their
    own maker): the reference's OWN load_rigid_body (particle_system.py:421-447), executed under the shim, now voxelises with
    oracle/voxel_check -- and the committed body fixtures come out bit for bit (round 6 regenerated all five completely, every
    array of every stage identical; this test re-executes the constructor + initialize() of two of them)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import importlib.util, json, copy, sys
import numpy as np
spec = importlib.util.spec_from_file_location("gen_golden", "oracle/gen_golden.py")
gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
assert "sph_taichi_amd.voxelizer" not in sys.modules
for name in ("ref_rigid_bodies", "ref_high_faces_rigid"):
    z = np.load(f"tests/golden/{name}.npz")
    sd = json.loads(str(z["scene"]))
    res = gg.flatten(gg.run_reference(copy.deepcopy(sd), 0))
    assert "sph_taichi_amd.voxelizer" not in sys.modules, "the shim pulled the product's voxeliser in"
    bad = [k for k in res if not np.array_equal(res[k], z[k])]
    assert not bad and len(res) >= 28, (name, bad, len(res))
    solid = int((z["initial/material"] == 0).sum())
    print(name, "ok", len(res), "arrays,", solid, "solid particles")
'''
    p = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert p.stdout.decode().count(" ok ") == 2
