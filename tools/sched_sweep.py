#!/usr/bin/env python
"""The SAME kernel sources under other instruction-scheduling strategies of the AMDGPU back end (round 6).

The two brick sweeps are bound by VALU issue (density: valu_busy 0.92 / 1.0) and by dependent LDS / vector-memory latency
(force: 0.80 / 0.75) -- exactly what a scheduler can move -- and five rounds had only ever used the default strategy.

  python tools/sched_sweep.py build            (CPU: hipcc each variant into sph_taichi_amd/libsph_hip_x-<tag>.so, resources table)
  python tools/sched_sweep.py run [--reps 2]   (GPU: bench.py per variant, alternating; JSON to gpurun_out/<out>)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sph_taichi_amd import build as B  # noqa: E402

VARIANTS = {
    "x-maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "x-memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "x-iterilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "x-minreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
    "x-maxocc": ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"],
    "x-bias100": ["-mllvm", "-amdgpu-schedule-metric-bias=100"],
    "x-bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "x-trackers": ["-mllvm", "-amdgpu-use-amdgpu-trackers"],
    "x-nopostra": ["-mllvm", "-enable-post-misched=0"],
}


def lib_of(tag):
    return os.path.join(os.path.dirname(B.LIB), f"libsph_hip_{tag}.so")


def do_build(tags):
    for tag in tags:
        out = lib_of(tag)
        cmd = [B.hipcc()] + B.FLAGS + VARIANTS[tag] + ["-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES]
        r = subprocess.run(cmd, cwd=B.CSRC, capture_output=True, text=True)
        print(tag, "rc", r.returncode, os.path.getsize(out) if os.path.exists(out) else None, r.stderr[-300:] if r.returncode else "")


def do_run(tags, reps, out, extra):
    res = {}
    order = ["default"] + tags
    for rep in range(reps):
        for tag in order:
            env = dict(os.environ)
            if tag != "default":
                if not os.path.exists(lib_of(tag)):
                    continue
                env["SPH_HIP_LIB_VARIANT"] = tag
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-steps", "0", "--with-bodies", "0"] + extra,
                               env=env, capture_output=True, text=True, timeout=400)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                row = {"rest_ms": d["ms_per_step"], "rest": d["breakdown_ms"], "settled_ms": d["settled"]["ms_per_step"], "settled": d["settled"]["breakdown_ms"]}
            except Exception as e:  # keep going: one broken variant must not lose the table
                row = {"error": repr(e), "stderr": r.stderr[-400:]}
            res.setdefault(tag, []).append(row)
            print(tag, rep, json.dumps(row), flush=True)
    with open(out, "w") as fh:
        json.dump({"flags": {k: " ".join(v) for k, v in VARIANTS.items()}, "runs": res}, fh, indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    tags = [t for t in a if t.startswith("x-")] or list(VARIANTS)
    if a and a[0] == "build":
        do_build(tags)
    elif a and a[0] == "run":
        reps = int(a[a.index("--reps") + 1]) if "--reps" in a else 2
        out = a[a.index("--out") + 1] if "--out" in a else os.path.join(ROOT, "gpurun_out", "sched_sweep.json")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        do_run(tags, reps, out, [])
    else:
        print(__doc__)
