"""Slab decomposition.  CPU: cut planes, scene subsetting, and the
TorchTransport exchange over gloo with world_size 2 and 3.  GPU (one device):
P logical slabs in one process, and 2 gloo ranks sharing the GPU, against the
single-domain run -- same particles, same trajectories."""
import copy
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import scenes
from sph_taichi_amd import scene as scene_mod
from sph_taichi_amd.config_builder import SimConfig

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(mode, world, out, extra=(), env=None, timeout=600):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), mode, str(r), str(world),
                               str(port), out, *extra], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=None if env is None else dict(os.environ, **env))
             for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0].decode())
    except subprocess.TimeoutExpired:
        for p in procs:             # exactly the processes started here
            p.kill()
        logs += [p.communicate()[0].decode() for p in procs[len(logs):]]
        raise AssertionError(f"{mode} workers hung (> {timeout} s):\n" + "\n".join(logs)[-4000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    return logs


def test_slab_cuts_balance_and_width():
    hist = np.zeros(400, dtype=np.int64)
    hist[1:257] = 165 * 165 * 2                      # C4: 512 particle planes in cells 1..256
    cuts = scene_mod.slab_cuts(hist, 8)
    assert cuts[0] == 0 and cuts[-1] == 400 and all(b - a >= 3 for a, b in zip(cuts, cuts[1:]))
    own = [hist[a:b].sum() for a, b in zip(cuts, cuts[1:])]
    assert max(own) - min(own) <= hist.max()           # balanced to one cell layer
    with pytest.raises(ValueError):
        scene_mod.slab_cuts(np.ones(5), 4)


def test_scene_subsets_partition_the_scene():
    sd = scenes.fluid_with_rigid_blocks()
    cfg, full = scenes.build(sd)
    hist = scene_mod.x_layer_histogram(cfg)
    assert hist.sum() == full.particle_max_num
    layers = scene_mod.x_layer_of(full.arrays["x"][:, 0], full.geom.grid_size, int(full.geom.grid_num[0]))
    assert np.array_equal(np.bincount(layers, minlength=len(hist)), hist)
    cuts = scene_mod.slab_cuts(hist, 3)
    nx = int(full.geom.grid_num[0])
    pids = []
    for r in range(3):
        lo, hi = cuts[r], cuts[r + 1]
        f = lambda xs: (scene_mod.x_layer_of(xs, full.geom.grid_size, nx) >= lo) & (scene_mod.x_layer_of(xs, full.geom.grid_size, nx) < hi)
        sub = scene_mod.build_scene(SimConfig(config=copy.deepcopy(sd)), x_filter=f)
        a = sub.arrays
        for k in ("x", "v", "density", "material", "is_dynamic", "object_id", "m"):
            assert np.array_equal(a[k], full.arrays[k][a["pid"]]), k
        pids.append(a["pid"])
    assert np.array_equal(np.sort(np.concatenate(pids)), np.arange(full.particle_max_num))


@pytest.mark.parametrize("world", [2, 3])
def test_transport_gloo(world, tmp_path):
    out = str(tmp_path / "res.txt")
    _spawn("transport", world, out)
    assert open(out).read() == "ok"


def test_bench_scenes_are_the_named_workloads():
    """BASELINE.json config 5 and the tiled weak-scaling family, as `bench.py --gpus N` builds them: particle
    counts, grids, and slabs of ~N / world particles each from the layer histogram (CPU: no particle array is built)."""
    from sph_taichi_amd.distributed import c4_dambreak_scene, slab_bench_scene, HALO
    sd, n = c4_dambreak_scene()
    assert n == 512 * 165 * 165 == 13_939_200
    cfg = SimConfig(config=copy.deepcopy(sd))
    geom = scene_mod.Geometry(cfg)
    assert tuple(int(v) for v in geom.grid_num) == (400, 100, 85)
    hist = scene_mod.x_layer_histogram(cfg, base_dir=None)
    assert int(hist.sum()) == n and int(np.count_nonzero(hist)) == 256         # 512 particle planes of spacing d = h / 2 from x = h on: two per cell layer
    for world in (2, 4, 8):
        cuts = scene_mod.slab_cuts(hist, world, min_width=HALO + 1)
        own = [int(hist[a:b].sum()) for a, b in zip(cuts, cuts[1:])]
        assert sum(own) == n and max(own) <= n / world + hist.max(), (world, own)
    for world in (1, 2, 8):
        sd, n = slab_bench_scene(world)
        assert n == 246 * world * 74 * 96 and sd["Configuration"]["domainEnd"] == [5.0 * world, 3.0, 2.0]
    sd, n = c4_dambreak_scene(0.2)
    assert n == 102 * 33 * 33


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_transport_negotiation_falls_back_collectively(bad_rank, tmp_path):
    """when ONE rank cannot use the native transport (here: its librccl does not open) every rank must
    take the torch transport together -- the negotiation agrees stage by stage, so nobody is left alone inside a
    collective.  CPU, gloo, world 2; either rank may be the broken one."""
    out = str(tmp_path / "res.txt")
    _spawn("negotiate", 2, out, extra=(str(bad_rank),), timeout=120)
    assert open(out).read() == "ok"


# ---------------------------------------------------------------------------
def _single_domain(sd, steps):
    cfg, sc = scenes.build(sd)
    ps, solver = scenes.make_ps(sd)
    solver.initialize()
    solver.step(steps)
    out = {k: scenes.ps_by_pid(ps, k) for k in ("x", "v", "density")}
    ps.close()
    return out, sc.particle_max_num


def _slab_scenes():
    a = scenes.fluid_only(counts=(20, 10, 8), start=(0.1, 0.1, 0.1), velocity=(1.5, -1.0, 0.0))   # drifts across the cuts
    b = scenes.fluid_with_rigid_blocks()
    c = scenes.crowded_fluid()      # bricks cut by the target limit: the boundary sweeps must keep the density sweep's cut
    return [a, b, c]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("which", [0, 1, 2])
def test_local_slabs_match_single_domain(world, which):
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd = _slab_scenes()[which]
    steps = 25
    ref, n = _single_domain(sd, steps)
    solvers = [SlabSolver(sd, r, world, device=0) for r in range(world)]
    run_local_slabs(solvers, 1, initialize=True)
    run_local_slabs(solvers, steps)
    x = gather_by_pid(solvers, "x", n)
    assert not np.isnan(x).any(), "a particle is owned by no rank"
    owned_total = sum(s.owned_range[1] for s in solvers)
    assert owned_total == n, "a particle is owned by two ranks"
    assert sum(s.stats["sent"] for s in solvers) > 0
    assert scenes.rel_l2(x, ref["x"]) <= 2e-6
    assert scenes.rel_l2(gather_by_pid(solvers, "v", n), ref["v"]) <= 2e-4
    for s in solvers:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1])
def test_local_slabs_restart_from_a_state(which):
    """SlabSolver(state=...): a slab job restarted from positions / velocities by persistent id (what gather_by_pid of an
    earlier run returns) continues like the single domain restarted from the same state -- cuts planned on the restart
    positions, every rank keeping what lies in its layers NOW."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd = _slab_scenes()[which]
    dev, n = _single_domain(sd, 30)
    state = {"x": dev["x"], "v": dev["v"]}
    ps, solver = scenes.make_ps(sd, arrays=state)
    solver.initialize()
    solver.step(12)
    ref = scenes.ps_by_pid(ps, "x")
    ps.close()
    solvers = [SlabSolver(sd, r, 3, device=0, state=state) for r in range(3)]
    run_local_slabs(solvers, 0, initialize=True)
    assert sum(s.owned_range[1] for s in solvers) == n
    run_local_slabs(solvers, 12)
    x = gather_by_pid(solvers, "x", n)
    for s in solvers:
        s.close()
    assert not np.isnan(x).any()
    assert scenes.rel_l2(x, ref) <= 2e-6


@pytest.mark.gpu
def test_restart_with_dynamic_bodies_keeps_the_rest_centre_of_mass(tmp_path):
    """a restart (`state` = positions / velocities by persistent id) keeps x_0 at the scene file's rest
    shape -- and must take the bodies' REST centre of mass from x_0 too (the reference computes it while x_0 == x,
    sph_base.py:80-90, 182-192), not from the displaced restart positions: shape matching pairs x_0 - rest_cm with x - cm
    (sph_base.py:206-218), so a rest centre taken from x makes every body jump at its first solve_constraints.  Single domain
    and two slabs: rigid_rest_cm bit-equal to the uninterrupted run's, and the continued trajectory follows it."""
    import copy
    from sph_taichi_amd import ParticleSystem, SimConfig
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd = scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"), fluid_velocity=(0.8, -1.0, 0.0))
    cfg, sc = scenes.build(sd)
    n = sc.particle_max_num
    rigid = (sc.arrays["material"] == 0) & (sc.arrays["is_dynamic"] == 1)
    ps, solver = scenes.make_ps(sd)
    solver.initialize()
    rest_cm = ps.rigid_rest_cm.to_numpy().copy()
    solver.step(30)
    state = {"x": scenes.ps_by_pid(ps, "x"), "v": scenes.ps_by_pid(ps, "v")}
    moved = np.abs(state["x"][rigid] - sc.arrays["x"][rigid]).max()
    assert moved > 0.01, "the bodies have not moved: the restart would not notice a rest centre taken from x"
    solver.step(12)
    ref = scenes.ps_by_pid(ps, "x")
    ps.close()
    # single domain, restarted
    ps2 = ParticleSystem(SimConfig(config=copy.deepcopy(sd)), state=state)
    solver2 = ps2.build_solver()
    solver2.initialize()
    got = ps2.rigid_rest_cm.to_numpy()
    for oid in (1, 2):
        assert np.array_equal(got[oid], rest_cm[oid]), f"body {oid}: rest centre {got[oid]} != {rest_cm[oid]} of the uninterrupted run"
    solver2.step(12)
    x2 = scenes.ps_by_pid(ps2, "x")
    ps2.close()
    assert scenes.rel_l2(x2, ref) <= 2e-6 and scenes.rel_l2(x2[rigid], ref[rigid]) <= 2e-6
    # two slabs, restarted
    solvers = [SlabSolver(sd, r, 2, device=0, state=state) for r in range(2)]
    run_local_slabs(solvers, 0, initialize=True)
    for s in solvers:
        got = s.ps.rigid_rest_cm.to_numpy()
        for oid in (1, 2):
            assert np.array_equal(got[oid], rest_cm[oid])
    run_local_slabs(solvers, 12)
    x3 = gather_by_pid(solvers, "x", n)
    for s in solvers:
        s.close()
    assert not np.isnan(x3).any()
    assert scenes.rel_l2(x3, ref) <= 2e-6 and scenes.rel_l2(x3[rigid], ref[rigid]) <= 2e-6


@pytest.mark.gpu
def test_slab_rank_checkpoint_does_not_trip_over_consumed_accelerations(tmp_path):
    """after a fused sph_slab_forces the interior accelerations are never materialised and their download is
    refused; save_state() of that rank's ParticleSystem must still write a checkpoint (the field is dead across steps)."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs
    sd = _slab_scenes()[0]
    solvers = [SlabSolver(sd, r, 2, device=0) for r in range(2)]
    run_local_slabs(solvers, 0, initialize=True)
    run_local_slabs(solvers, 3)
    ck = str(tmp_path / "rank0.npz")
    solvers[0].ps.save_state(ck)
    z = np.load(ck)
    assert z["x"].shape[0] == solvers[0].ps.count() and z["acceleration"].shape == z["x"].shape and np.isfinite(z["acceleration"]).all()
    for s in solvers:
        s.close()


@pytest.mark.gpu
def test_layer_offsets_of_an_empty_record_set_are_zero():
    """Since round 6 a slab rank's layer offsets ride in the sort: its place kernel writes them and a stamp into mapped host
    memory and sph_layer_offsets_end spins on the stamp.  A sort over NO records launches no place kernel: the call must answer
    with zeros at once -- not spin, and not hand back the mapped words of the step before."""
    import ctypes as C
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs
    sd = _slab_scenes()[0]
    solvers = [SlabSolver(sd, r, 2, device=0) for r in range(2)]
    run_local_slabs(solvers, 0, initialize=True)
    run_local_slabs(solvers, 2)
    s = solvers[0]
    assert s.off[3] > 0                                   # the step before left non-zero counts in the mapped words
    layers = s._layers()
    nl = len(layers)
    s.ps._call("sph_slab_advance", 0, 0, None, 0, None, 0, (C.c_int32 * nl)(*layers), nl, 0)   # keep nothing, receive nothing
    out = (C.c_int32 * nl)(*([-1] * nl))
    s.ps._call("sph_layer_offsets_end", out, nl)
    assert list(out) == [0] * nl
    for s in solvers:
        s.close()


@pytest.mark.gpu
def test_interior_accelerations_are_refused_while_they_are_not_materialised():
    """in slab mode the interior force sweep integrates its targets in its finish and does not write their
    accelerations out; a download used to return stale values silently.  Now it fails with a message, and a stand-alone
    force computation makes the field readable again."""
    from sph_taichi_amd import _lib
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs
    sd = _slab_scenes()[0]
    solvers = [SlabSolver(sd, r, 2, device=0) for r in range(2)]
    run_local_slabs(solvers, 0, initialize=True)
    run_local_slabs(solvers, 3)
    ps = solvers[0].ps
    if ps.get_option(_lib.OPT_UNIFORM_FLUID_STATE) == 1:      # the fused interior advect ran: the field is partial
        with pytest.raises(_lib.SphError, match="acceleration"):
            ps.acceleration.to_numpy()
        ps._call("sph_compute_non_pressure_forces")
        ps._call("sph_compute_pressure_forces")
    a = solvers[0].owned(("pid", "acceleration"))["acceleration"]     # (the outer ghost layer has no density: only owned rows mean anything)
    assert a.shape[0] == solvers[0].owned_range[1] and np.isfinite(a).all() and np.abs(a).max() > 0.0
    for s in solvers:
        s.close()


def test_plan_recut_moves_one_layer_towards_balance():
    from sph_taichi_amd.distributed import plan_recut
    from sph_taichi_amd.scene import slab_cuts
    hist = np.zeros(60, dtype=np.int64)
    hist[4:28] = 100                                  # the fluid sits in the left half of the tank
    for world, halo in ((2, 2), (3, 2), (4, 3)):
        target = slab_cuts(hist, world, min_width=halo + 1)
        cuts = [0] + [60 - (world - i) * (halo + 2) for i in range(1, world)] + [60]   # everything piled up on the right
        for it in range(200):
            new = plan_recut(cuts, hist, world, halo)
            assert new[0] == 0 and new[-1] == 60
            assert all(abs(a - b) <= 1 for a, b in zip(new, cuts))
            assert all(new[i + 1] - new[i] >= halo + 1 for i in range(world))
            for i in range(1, world):                 # a cut only moves into a slab that can pack one layer deeper
                if new[i] > cuts[i]:
                    assert cuts[i + 1] - cuts[i] >= halo + 2
                if new[i] < cuts[i]:
                    assert cuts[i] - cuts[i - 1] >= halo + 2
            if new == cuts:
                break
            cuts = new
        assert cuts == list(target), (cuts, target)
    # the allocation caps a slab's width: slab 0 may not grow beyond 12 layers
    cuts = [0, 10, 60]
    hist2 = np.zeros(60, dtype=np.int64)
    hist2[0:50] = 10
    for _ in range(10):
        cuts = plan_recut(cuts, hist2, 2, 2, width_cap=[12, 60])
    assert cuts == [0, 12, 60]
    # balanced already: nothing moves
    assert plan_recut(list(slab_cuts(hist, 3, min_width=3)), hist, 3, 2) == list(slab_cuts(hist, 3, min_width=3))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fluid-2", "fluid-3", "bodies-2", "dfsph-2"])
def test_local_slabs_recut(case, tmp_path):
    """Re-cut while running: the slabs start badly balanced, every other step each cut plane moves one
    layer towards balance -- ownership changes hands through the running exchange, and the trajectory stays the
    single-domain one."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid, plan_recut
    from sph_taichi_amd import scene as _scene
    from sph_taichi_amd.config_builder import SimConfig
    which, world = case.split("-")
    world = int(world)
    sd = {"fluid": lambda: _slab_scenes()[0], "dfsph": _dfsph_slab_scene,
          "bodies": lambda: scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"), fluid_velocity=(0.8, -1.0, 0.0))}[which]()
    steps = 24 if which != "dfsph" else 8
    ref, n = _single_domain(sd, steps)
    cfg = SimConfig(config=copy.deepcopy(sd))
    hist = _scene.x_layer_histogram(cfg, base_dir=None)
    halo = 3 if which == "bodies" else 2
    balanced = list(_scene.slab_cuts(hist, world, min_width=halo + 1))
    nx = len(hist)
    filled = np.nonzero(hist)[0]
    # skewed start: every interior cut 4 layers right of the balanced one (clipped so that every slab is valid)
    cuts = [0] + [min(c + 4, nx - (world - i) * (halo + 1)) for i, c in enumerate(balanced[1:-1], 1)] + [nx]
    assert cuts != balanced and filled.size > 0
    solvers = [SlabSolver(sd, r, world, device=0, cuts=cuts, recut_every=2) for r in range(world)]
    run_local_slabs(solvers, 1, initialize=True)
    run_local_slabs(solvers, steps)
    assert sum(s.stats.get("recuts", 0) for s in solvers) >= 2, "no cut moved"
    assert all(s.cuts == solvers[0].cuts for s in solvers) and solvers[0].cuts != cuts
    assert sum(abs(a - b) for a, b in zip(solvers[0].cuts, balanced)) < sum(abs(a - b) for a, b in zip(cuts, balanced))
    for r, s in enumerate(solvers):
        assert (s.x_lo, s.x_hi) == (s.cuts[r], s.cuts[r + 1])
    x = gather_by_pid(solvers, "x", n)
    assert not np.isnan(x).any(), "a particle is owned by no rank"
    assert sum(s.owned_range[1] for s in solvers) == n, "a particle is owned by two ranks"
    assert scenes.rel_l2(x, ref["x"]) <= (2e-6 if which != "dfsph" else 1e-4)
    assert scenes.rel_l2(gather_by_pid(solvers, "v", n), ref["v"]) <= (2e-4 if which != "dfsph" else 1e-2)
    for s in solvers:
        s.close()


@pytest.mark.gpu
def test_slab_window_is_bounded_by_the_allocation():
    from sph_taichi_amd.distributed import SlabSolver
    from sph_taichi_amd import _lib
    s = SlabSolver(_slab_scenes()[0], 0, 2, device=0, recut_every=5, nx_slack=2)
    w = s.x_hi - s.x_lo
    s.ps.set_slab_window(s.x_lo, s.x_hi + 2)                  # the slack
    assert s.ps._local_grid_num[0] == w + 2 + 2 * s.halo
    with pytest.raises(_lib.SphError, match="allocation"):
        s.ps.set_slab_window(s.x_lo, s.x_hi + 3)
    assert s.ps._local_grid_num[0] == w + 2 + 2 * s.halo      # a refused window changes nothing
    s.close()


@pytest.mark.gpu
def test_a_slab_rank_beyond_the_32_bit_list_offsets_is_refused():
    """The brick sweeps address the neighbour lists with 32-bit byte offsets (24 groups of 2^k >= 8 * capacity bytes): a
    context whose capacity does not fit (> 16.7 M particles) allocates no lists and runs the per-particle cell walk; a
    SLAB rank cannot (its boundary / interior launches are brick launches) and says so instead of computing garbage."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs
    from sph_taichi_amd import _lib
    sd = _slab_scenes()[0]
    n = scenes.build(sd)[1].particle_max_num
    s = SlabSolver(sd, 0, 1, device=0, capacity_factor=(1 << 24) / n + 64.0)
    assert s.capacity > (1 << 24)
    run_local_slabs([s], 1, initialize=True)
    with pytest.raises(_lib.SphError, match="more slabs"):
        run_local_slabs([s], 1)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_local_slabs_shape_matched_bodies(world, tmp_path):
    """Dynamic RigidBodies straddling the cut planes: per-rank sums + all-reduce reproduce the single-domain
    shape matching (sph_base.py:200-260); HALO = 3 keeps the moving boundary volumes of the ghosts exact."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd = scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"), fluid_velocity=(0.8, -1.0, 0.0))
    steps = 40
    ref, n = _single_domain(sd, steps)
    solvers = [SlabSolver(sd, r, world, device=0) for r in range(world)]
    assert solvers[0].halo == 3 and solvers[0].dynamic_bodies == [1, 2]
    cfg, sc = scenes.build(sd)
    layers = (sc.arrays["x"][sc.arrays["object_id"] == 2, 0] / np.float32(0.04)).astype(int)
    assert any(layers.min() < c <= layers.max() for c in solvers[0].cuts[1:-1]), "no body straddles a cut"
    run_local_slabs(solvers, 1, initialize=True)
    run_local_slabs(solvers, steps)
    x = gather_by_pid(solvers, "x", n)
    assert not np.isnan(x).any(), "a particle is owned by no rank"
    assert sum(s.owned_range[1] for s in solvers) == n
    assert scenes.rel_l2(x, ref["x"]) <= 2e-6
    rigid = (sc.arrays["material"] == 0) & (sc.arrays["is_dynamic"] == 1)
    assert scenes.rel_l2(x[rigid], ref["x"][rigid]) <= 2e-6
    assert scenes.rel_l2(gather_by_pid(solvers, "v", n), ref["v"]) <= 2e-4
    for s in solvers:
        s.close()


def _dfsph_slab_scene():
    """A fluid block thrown across the cut planes onto a static slab, under DFSPH (both solvers iterate)."""
    sd = scenes.fluid_with_rigid_blocks(fluid_counts=(16, 10, 8), static_counts=(20, 2, 12), dyn_counts=(4, 4, 4))
    sd["RigidBlocks"] = sd["RigidBlocks"][:1]
    sd["FluidBlocks"][0]["velocity"] = [1.5, -2.0, 0.0]
    return scenes.as_dfsph(sd, dt=0.002)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_local_slabs_dfsph(world):
    """DFSPH across slabs: ghost velocities refreshed after every Jacobi sweep, density error summed over ranks =>
    the same iteration counts and trajectory as the single-domain solver."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd = _dfsph_slab_scene()
    steps = 12
    cfg, sc = scenes.build(sd)
    ps, solver = scenes.make_ps(sd)
    solver.initialize()
    its = []
    for _ in range(steps):
        solver.step(1)
        st = solver.stats()
        its.append((st["iterations_v"], st["iterations"]))
    ref = {k: scenes.ps_by_pid(ps, k) for k in ("x", "v")}
    ps.close()
    assert sum(a + b for a, b in its) > 0, "the solvers never iterated: the scene does not test the refresh"
    solvers = [SlabSolver(sd, r, world, device=0) for r in range(world)]
    run_local_slabs(solvers, 1, initialize=True)
    got = []
    for _ in range(steps):
        run_local_slabs(solvers, 1)
        got.append(solvers[0].dfsph_iterations)
        assert all(s.dfsph_iterations == got[-1] for s in solvers)
    n = sc.particle_max_num
    x = gather_by_pid(solvers, "x", n)
    assert not np.isnan(x).any() and sum(s.owned_range[1] for s in solvers) == n
    # the density error is a sum in a different order (per-rank partials): an iteration count may flip where the
    # average error sits on the threshold (the reference's own f32 atomic reduction is not deterministic either)
    assert got[:5] == its[:5], f"iteration counts differ from the single-domain run: {got} vs {its}"
    assert abs(sum(a + b for a, b in got) - sum(a + b for a, b in its)) <= 2, (got, its)
    err = scenes.rel_l2(x, ref["x"])
    assert err <= 1e-4, err
    assert scenes.rel_l2(gather_by_pid(solvers, "v", n), ref["v"]) <= 5e-3
    assert sum(s.stats["sent"] for s in solvers) > 0
    for s in solvers:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_local_slabs_dfsph_with_dynamic_solids(world, tmp_path):
    """DFSPH across slabs with shape-matched bodies straddling the cuts (HALO 3, coupling reactions of the pressure
    solver, 16-sum all-reduce per body)."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd = scenes.as_dfsph(scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"), fluid_velocity=(0.8, -1.0, 0.0),
                                                        body_velocities=((0.3, -4.0, 0.2), (-0.25, -4.0, 0.35))), dt=0.002)
    steps = 16
    cfg, sc = scenes.build(sd)
    ps, solver = scenes.make_ps(sd)
    solver.initialize(); solver.step(steps)
    ref = {k: scenes.ps_by_pid(ps, k) for k in ("x", "v")}
    ps.close()
    solvers = [SlabSolver(sd, r, world, device=0) for r in range(world)]
    assert solvers[0].halo == 3 and solvers[0].dynamic_bodies == [1, 2]
    run_local_slabs(solvers, 1, initialize=True)
    run_local_slabs(solvers, steps)
    n = sc.particle_max_num
    x = gather_by_pid(solvers, "x", n)
    assert not np.isnan(x).any() and sum(s.owned_range[1] for s in solvers) == n
    rigid = (sc.arrays["material"] == 0) & (sc.arrays["is_dynamic"] == 1)
    assert np.abs(ref["v"][rigid, 1] + 4.0).max() > 0.05, "the bodies never felt the fluid: nothing coupled"
    assert scenes.rel_l2(x, ref["x"]) <= 1e-4
    assert scenes.rel_l2(x[rigid], ref["x"][rigid]) <= 1e-4
    for s in solvers:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["fluid", "bodies", "dfsph", "fluid_recut"])
def test_two_gloo_ranks_on_one_gpu(tmp_path, which):
    sd = {"fluid": lambda: _slab_scenes()[0], "dfsph": _dfsph_slab_scene, "fluid_recut": lambda: _slab_scenes()[0],
          "bodies": lambda: scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"))}[which]()
    steps = 20 if which != "dfsph" else 8
    ref, n = _single_domain(sd, steps)
    scene_file = str(tmp_path / "scene.json")
    json.dump(sd, open(scene_file, "w"))
    out = str(tmp_path / "res.npz")
    extra = (scene_file, str(steps))
    if which == "fluid_recut":      # badly cut at the start, re-cut every other step (histogram all-reduce over gloo)
        from sph_taichi_amd import scene as _scene
        from sph_taichi_amd.config_builder import SimConfig
        hist = _scene.x_layer_histogram(SimConfig(config=copy.deepcopy(sd)), base_dir=None)
        bal = list(_scene.slab_cuts(hist, 2, min_width=3))
        start = [0, min(bal[1] + 4, len(hist) - 3), len(hist)]
        extra += ("2", json.dumps(start))
    _spawn("slabs", 2, out, extra=extra)
    z = np.load(out)
    if which == "fluid_recut":
        assert int(z["recuts"]) >= 1 and abs(int(z["cuts"][1]) - bal[1]) < abs(start[1] - bal[1])
    assert np.array_equal(np.sort(z["pid"]), np.arange(n))
    x = np.empty_like(ref["x"])
    x[z["pid"]] = z["x"]
    assert scenes.rel_l2(x, ref["x"]) <= (2e-6 if which != "dfsph" else 1e-4)


@pytest.mark.gpu
def test_rccl_one_rank_transport_and_slab_solver(tmp_path):
    """The RCCL code path on the hardware there is: one rank on backend "nccl" with device tensors end to end --
    TorchTransport's whole protocol with the rank as its own neighbour (send/recv to self), the 16-double
    all-reduce of the shape-matched bodies, the histogram all-reduce of the re-cut events and the conservation
    guard's count all-reduce inside a stepping SlabSolver -- against the single-domain trajectory."""
    sd = scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"))
    steps = 12
    ref, n = _single_domain(sd, steps)
    scene_file = str(tmp_path / "scene.json")
    json.dump(sd, open(scene_file, "w"))
    out = str(tmp_path / "res.npz")
    port = _free_port()
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), "rccl1", "0", "1", str(port), out,
                          scene_file, str(steps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        log = p.communicate(timeout=240)[0].decode()
    except subprocess.TimeoutExpired:
        p.kill()
        raise AssertionError("RCCL one-rank worker hung:\n" + p.communicate()[0].decode()[-3000:])
    assert p.returncode == 0, log[-3000:]
    z = np.load(out)
    assert str(z["backend"]) == "nccl" and int(z["ok"]) == 1, log[-2000:]
    assert np.array_equal(np.sort(z["pid"]), np.arange(n))
    x = np.empty_like(ref["x"])
    x[z["pid"]] = z["x"]
    assert scenes.rel_l2(x, ref["x"]) <= 2e-6


@pytest.mark.gpu
def test_native_rccl_exchange_behind_the_c_abi(tmp_path):
    """SURVEY 8(b) sph_exchange_halo / sph_migrate as C entry points (csrc/sph_comm.hip): communicator from a unique id,
    counts announced ahead, ncclSend / ncclRecv groups enqueued on the communication stream, fixed-size swap, f64 and
    i64 all-reduce -- executed with ONE rank as its own neighbour (the hardware there is), then a SlabSolver with
    shape-matched bodies, re-cut events and the conservation guard running over NativeTransport, against the single
    domain."""
    sd = scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"))
    steps = 12
    ref, n = _single_domain(sd, steps)
    scene_file = str(tmp_path / "scene.json")
    json.dump(sd, open(scene_file, "w"))
    out = str(tmp_path / "res.npz")
    port = _free_port()
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), "native1", "0", "1", str(port), out,
                          scene_file, str(steps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        log = p.communicate(timeout=240)[0].decode()
    except subprocess.TimeoutExpired:
        p.kill()
        raise AssertionError("native RCCL worker hung:\n" + p.communicate()[0].decode()[-3000:])
    assert p.returncode == 0, log[-3000:]
    z = np.load(out)
    assert str(z["backend"]) == "native-rccl" and int(z["ok"]) == 1, log[-2000:]
    assert np.array_equal(np.sort(z["pid"]), np.arange(n))
    x = np.empty_like(ref["x"])
    x[z["pid"]] = z["x"]
    assert scenes.rel_l2(x, ref["x"]) <= 2e-6


def _fake_rccl_env(slot_bytes=65536):
    """SPH_RCCL_LIB -> the stand-in for librccl (tests/fake_rccl/): several PROCESSES on one GPU run the exchange of
    csrc/sph_comm.hip.  Small ring slots, so that record payloads span several of them; short timeouts, so that a
    protocol bug is a failed test within a minute, never a hung box."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sph_fake_rccl_build", os.path.join(HERE, "fake_rccl", "build.py"))
    fake_build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fake_build)
    return {"SPH_RCCL_LIB": fake_build.build(), "FAKE_RCCL_SLOT_BYTES": str(slot_bytes), "FAKE_RCCL_SLOTS": "3",
            "FAKE_RCCL_TIMEOUT_S": "25"}


NATIVE_CASES = {
    # name: (scene, world, steps, worker options, x tolerance)
    "fluid-2": ("fluid", 2, 24, {}, 2e-6),
    "fluid-3": ("fluid", 3, 24, {"check_every": 3}, 2e-6),                      # + the conservation guard's all-reduce
    "fluid-3-slow-rank": ("fluid", 3, 16, {"delay_rank": 1, "delay_ms": 25}, 2e-6),
    "bodies-2": ("bodies", 2, 20, {"check_every": 5}, 2e-6),                   # 16-sum all-reduce per body, HALO 3, synchronous packers
    "bodies-3-recut": ("bodies", 3, 16, {"recut": 2, "skew": 3}, 2e-6),
    "dfsph-2": ("dfsph", 2, 8, {}, 1e-4),                                      # ghost-velocity swap per Jacobi sweep, error all-reduce
    "fluid-3-recut": ("fluid", 3, 24, {"recut": 2, "skew": 4}, 2e-6),          # histogram all-reduce, cuts travel
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(NATIVE_CASES))
def test_native_exchange_between_processes(case, tmp_path):
    """VERDICT r03 "missing" #1: the exchange behind the C ABI (sph_slab_announce / _incoming / _exchange, sph_comm_swap,
    sph_comm_all_reduce) had only ever run as ONE rank talking to itself.  Here 2 and 3 PROCESSES share the GPU and run
    SlabSolver over NativeTransport through tests/fake_rccl (RCCL refuses two ranks on a device): counts of step n+1
    announced on the communication stream while step n's payload group is pending, payloads enqueued behind the packers'
    event, the insert + sort behind the exchange's event -- with fluid drifting across the cuts, shape-matched bodies,
    the DFSPH swaps, re-cut events, the conservation guard, and one deliberately slow rank.  Each against the
    single-domain trajectory at the bounds of the logical-slab tests."""
    which, world, steps, opt, tol = NATIVE_CASES[case]
    sd = {"fluid": lambda: _slab_scenes()[0], "dfsph": _dfsph_slab_scene,
          "bodies": lambda: scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"), fluid_velocity=(0.8, -1.0, 0.0))}[which]()
    ref, n = _single_domain(sd, steps)
    opt = dict(opt)
    start = bal = None
    if "skew" in opt:               # badly cut at the start: the re-cut has something to do
        from sph_taichi_amd import scene as _scene
        hist = _scene.x_layer_histogram(SimConfig(config=copy.deepcopy(sd)), base_dir=None)
        halo = 3 if which == "bodies" else 2
        bal = list(_scene.slab_cuts(hist, world, min_width=halo + 1))
        nx = len(hist)
        k = opt.pop("skew")
        start = [0] + [min(c + k, nx - (world - i) * (halo + 1)) for i, c in enumerate(bal[1:-1], 1)] + [nx]
        assert start != bal
        opt["cuts"] = start
    scene_file = str(tmp_path / "scene.json")
    json.dump(sd, open(scene_file, "w"))
    out = str(tmp_path / "res.npz")
    _spawn("native", world, out, extra=(scene_file, str(steps), json.dumps(opt)), env=_fake_rccl_env(), timeout=150)
    z = np.load(out)
    assert int(z["ok"]) == 1, "the transport-level checks (ragged exchange / swap / all-reduce between processes) failed"
    assert int(z["sent"]) > 0 and int(z["exchanges"]) >= steps
    assert np.array_equal(np.sort(z["pid"]), np.arange(n)), "a particle is owned by no rank or by two"
    x = np.empty_like(ref["x"])
    x[z["pid"]] = z["x"]
    assert scenes.rel_l2(x, ref["x"]) <= tol
    v = np.empty_like(ref["v"])
    v[z["pid"]] = z["v"]
    assert scenes.rel_l2(v, ref["v"]) <= (2e-4 if which != "dfsph" else 1e-2)
    if start is not None:
        assert int(z["recuts"]) >= 1, "no cut moved"
        cuts = [int(c) for c in z["cuts"]]
        assert sum(abs(a - b) for a, b in zip(cuts, bal)) < sum(abs(a - b) for a, b in zip(start, bal))


@pytest.mark.gpu
def test_native_exchange_reports_a_missing_peer_instead_of_hanging(tmp_path):
    """The stand-in's waits time out and latch an error: a rank whose neighbour never shows up fails with a message."""
    sd = _slab_scenes()[0]
    scene_file = str(tmp_path / "scene.json")
    json.dump(sd, open(scene_file, "w"))
    env = dict(_fake_rccl_env(), FAKE_RCCL_TIMEOUT_S="3")
    port = _free_port()
    # world = 2 but only rank 0's NativeTransport is created: gloo is initialised by both, rank 1 exits before the comm
    p = [subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), m, str(r), "2", str(port), str(tmp_path / "o.npz"),
                           scene_file, "2", "{}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **env))
         for r, m in ((0, "native"), (1, "native_absent"))]
    logs = []
    try:
        logs = [q.communicate(timeout=120)[0].decode() for q in p]
    except subprocess.TimeoutExpired:
        for q in p:
            q.kill()
        raise AssertionError("a missing peer hung the worker instead of raising")
    assert p[0].returncode != 0 and "fake_rccl" in logs[0] and "timeout" in logs[0], logs[0][-2000:]


def _bench_two_ranks(transport, launcher, extra=(), tmp_path=None, backend="gloo"):
    root = os.path.dirname(HERE)
    env = dict(os.environ, SPH_DIST_BACKEND=backend, SPH_C4_SCALE="0.2", SPH_TRANSPORT=transport)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    if transport == "native":
        env.update(_fake_rccl_env(slot_bytes=1 << 20))
    args = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--settled-after", "60",
            "--preheat-ms", "0", "--watchdog-s", "420", *extra]
    if launcher == "torchrun":      # as the driver launches N > 1: torch.distributed.run, one process per rank
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:                           # as the driver launches N = 1: plain python -- bench.py spawns its ranks itself
        cmd = [sys.executable] + args
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600, cwd=root)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.gpu
@pytest.mark.parametrize("transport,launcher", [("torch", "torchrun"), ("native", "self")])
def test_bench_two_ranks_on_one_gpu(transport, launcher, tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank) AND with no launcher at
    all (`python bench.py --gpus 2`: bench.py spawns its own ranks, VERDICT r04 "next" #2a) -- here with both ranks on the one
    GPU there is (SPH_DIST_BACKEND=gloo), once over the torch transport and once over NativeTransport through the librccl
    stand-in, negotiated collectively: the tiled weak-scaling line AND BASELINE.json's config 5 (`c4_dambreak`, shrunk by
    SPH_C4_SCALE so that it fits a test) with re-cut, a settled state, per-rank owned counts and conservation."""
    p, lines = _bench_two_ranks(transport, launcher)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["particles"] == 2 * 1_747_584
    assert d["config"]["transport"] == ("NativeTransport" if transport == "native" else "TorchTransport"), d["config"]
    assert d["config"]["backend"] == "gloo" and d["config"]["comm"]["world"] == 2 and d["config"]["comm"]["rank"] == 0
    assert sum(d["config"]["particles_owned_per_rank"]) == d["config"]["particles"]
    c4 = d["c4_dambreak"]
    assert "error" not in c4, c4
    assert c4["conserved"] and c4["recut_every"] == 10 and c4["from_rest"]["value"] > 0 and c4["settled"]["value"] > 0
    assert sum(c4["owned_end"]) == c4["particles"] and len(c4["owned_end"]) == 2 and c4["comm"]["world"] == 2
    assert c4["imbalance_end"] <= 1.25, c4
    if transport == "native":
        assert c4["transport"] == "NativeTransport" and c4["halo_device_ms"] >= 0.0 and "halo_device" in d["breakdown_ms"]
        assert "RCCL behind the C ABI" in c4["comm"]["library"]


@pytest.mark.gpu
def test_bench_with_the_control_plane_on_gloo_and_rccl_for_the_records_only(tmp_path):
    """SPH_DIST_BACKEND="cpu:gloo,cuda:nccl" (VERDICT r04 "weak" #11c: the nccl-backend job has TWO RCCL communicators, torch's
    and the library's): the control plane (negotiation, barriers, timing and owned-count reductions) runs on CPU tensors over
    gloo, torch's RCCL process group is created lazily -- i.e. never, unless the torch transport has to carry the records --
    and the library's communicator is the only one.  Two ranks on the one GPU: any CUDA collective through torch would die
    with "Duplicate GPU detected", so the run passing IS the proof that none was issued."""
    p, lines = _bench_two_ranks("native", "self", backend="cpu:gloo,cuda:nccl")
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "cpu:gloo,cuda:nccl" and d["config"]["transport"] == "NativeTransport"
    assert d["config"]["comm"]["world"] == 2 and d["value"] > 0 and "RCCL" in d["config"]["parallelism"]
    assert "error" not in d["c4_dambreak"] and d["c4_dambreak"]["conserved"]


@pytest.mark.gpu
def test_bench_keeps_the_tiled_line_when_the_c4_object_hangs(tmp_path):
    """ADVICE r04 medium / VERDICT r04 "next" #2c: the supplementary c4_dambreak object gets a wall-clock budget it cannot
    meet (the stand-in for a hang inside it): both ranks' watchdogs end the job, rank 0 prints the FINISHED tiled line with
    c4_dambreak = {error: watchdog, stage: ...}, exit code 0."""
    # (the shrunk c4 object finishes in a few hundred ms: it is given 40,000 settling steps and half a second)
    p, lines = _bench_two_ranks("torch", "self", extra=("--c4-budget-s", "0.5", "--settled-after", "40000"))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and sum(d["config"]["particles_owned_per_rank"]) == d["config"]["particles"]
    assert d["c4_dambreak"]["error"] == "watchdog" and d["c4_dambreak"]["stage"].startswith("c4_dambreak:"), d["c4_dambreak"]


@pytest.mark.gpu
def test_bench_hang_before_the_line_is_rc_124_and_names_the_stage(tmp_path):
    """A whole-job budget that ends inside the tiled run: no line is complete, so the job must end with rc != 0 and ONE
    JSON line naming the stage (never a silent hang until the driver's own timeout)."""
    p, lines = _bench_two_ranks("torch", "self", extra=("--watchdog-s", "25", "--steps", "200000"))
    assert p.returncode == 124, (p.returncode, p.stderr.decode()[-2000:])
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["error"] == "watchdog" and d["stage"] and d["n_gpus"] == 2, d


@pytest.mark.gpu
def test_conservation_guard_raises_when_a_particle_outruns_the_halo():
    """a particle that crosses more than one cell layer in a step is dropped (or duplicated) by the
    exchange; the guard must turn that into an error instead of a silently different fluid."""
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs
    sd = scenes.fluid_only(counts=(20, 10, 8), start=(0.1, 0.1, 0.1), velocity=(1.5, -1.0, 0.0))
    sd["Configuration"]["timeStepSize"] = 0.05      # 1.5 m/s * 0.05 s = 0.075 m = almost two cell layers (h = 0.04) per step
    solvers = [SlabSolver(sd, r, 2, device=0) for r in range(2)]
    run_local_slabs(solvers, 0, initialize=True)
    with pytest.raises(RuntimeError, match="conservation"):
        run_local_slabs(solvers, 6)
    for s in solvers:
        s.close()
