"""Build libsph_hip.so (hipcc, gfx950 only) in-tree next to this file."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["sph_api.hip", "sph_sort.hip", "sph_gather.hip", "sph_integrate.hip"]
HEADERS = [os.path.join(CSRC, "sph_internal.h"), os.path.join(_HERE, "..", "include", "sph_hip.h")]
LIB = os.path.join(_HERE, "libsph_hip.so")
# -fno-slp-vectorize: the SLP pass packs the x/y lanes of the distance test into v_pk_*_f32, which on gfx950
# costs extra v_mov + s_nop per pair test (measured: slower than plain v_sub/v_fma)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fno-slp-vectorize"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsph_hip.so cannot be built")
    return exe


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
