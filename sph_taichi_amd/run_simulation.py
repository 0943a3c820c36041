"""Headless counterpart of the reference's run_simulation.py (argument parsing,
substep loop and exporters of /root/reference/run_simulation.py:12-35, 79-113;
the GGUI window/camera/render code of :37-74, 82-94, 118 has no counterpart).

    python -m sph_taichi_amd.run_simulation --scene_file data/scenes/dragon_bath.json --frames 100

Per frame: `numberOfStepsPerRenderUpdate` solver steps; every int(0.016/dt)
frames optional ASCII-PLY particle export (`exportPly`) and OBJ rigid-body export
(`exportObj`), written where the reference writes them.  `--timing` prints the
per-phase HIP-event breakdown (sort / neighbour / force / integrate).
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np

from . import _lib
from .config_builder import SimConfig
from .particle_system import ParticleSystem


def write_ply_ascii(path: str, pos: np.ndarray):
    """Vertex-only ASCII PLY, the layout ti.tools.PLYWriter.export_frame_ascii produces."""
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment created by sph_taichi_amd\n")
        fh.write(f"element vertex {pos.shape[0]}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        np.savetxt(fh, pos, fmt="%.6f")


def main(argv=None):
    ap = argparse.ArgumentParser(description="SPH (MI355X)")
    ap.add_argument("--scene_file", default="", help="scene file")
    ap.add_argument("--frames", type=int, default=100, help="number of frames (the reference loops until the window closes)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--dump_npz", default="", help="write the final per-particle state here")
    ap.add_argument("--save_state", default="", help="write a restartable checkpoint (.npz) after the last frame")
    ap.add_argument("--load_state", default="", help="continue from a checkpoint written by --save_state")
    args = ap.parse_args(argv)

    scene_path = args.scene_file
    config = SimConfig(scene_file_path=scene_path)
    scene_name = scene_path.split("/")[-1].split(".")[0]
    substeps = config.get_cfg("numberOfStepsPerRenderUpdate")
    output_interval = int(0.016 / config.get_cfg("timeStepSize"))
    output_ply = config.get_cfg("exportPly")
    output_obj = config.get_cfg("exportObj")
    series_prefix = "{}_output/particle_object_{}.ply".format(scene_name, "{}")
    if output_ply or output_obj:
        os.makedirs(f"{scene_name}_output", exist_ok=True)

    scene_dir = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(scene_path)))) or "."
    ps = ParticleSystem(config, GGUI=False, device=args.device, scene_dir=scene_dir, verbose=True)
    solver = ps.build_solver()
    solver.initialize()
    if args.load_state:
        meta = ps.load_state(args.load_state)
        print(f"restored {args.load_state} (frames done: {int(meta.get('frames', 0))})")
    if args.timing:
        ps.set_option(_lib.OPT_TIMING, 1)

    cnt = cnt_ply = 0
    t0 = time.perf_counter()
    for _ in range(args.frames):
        solver.step(substeps)
        if cnt % output_interval == 0:
            if output_ply:
                obj_id = 0
                np_pos = ps.dump(obj_id=obj_id)["position"]
                write_ply_ascii(series_prefix.format(0).replace(".ply", f"_{cnt_ply:06}.ply"), np_pos)
            if output_obj:
                for r_body_id in ps.object_id_rigid_body:
                    with open(f"{scene_name}_output/obj_{r_body_id}_{cnt_ply:06}.obj", "w") as f:
                        f.write(ps.object_collection[r_body_id]["mesh"].export(file_type="obj"))
            cnt_ply += 1
        cnt += 1
    ps.sync()
    dt = time.perf_counter() - t0
    steps = args.frames * substeps
    report = {"scene": scene_name, "particles": ps.particle_max_num, "steps": steps,
              "steps_per_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4)}
    if args.timing:
        tm = _lib.SphTimings()
        ps._call("sph_get_timings", tm)
        k = max(int(tm.steps), 1)
        report["breakdown_ms"] = {"sort": tm.sort_ms / k, "neighbour": tm.neighbour_ms / k, "force": tm.force_ms / k,
                                  "integrate": tm.integrate_ms / k}
    print(json.dumps(report))
    if args.save_state:
        ps.save_state(args.save_state, frames=args.frames, scene=scene_name)
    if args.dump_npz:
        np.savez_compressed(args.dump_npz, **{f: getattr(ps, f).to_numpy() for f in
                                              ("object_id", "x", "v", "density", "pressure", "m_V", "pid")})
    ps.close()


if __name__ == "__main__":
    main()
