"""Build tests/fake_rccl/libfake_rccl.so (hipcc, gfx950): the TEST-ONLY stand-in for librccl that lets several processes
sharing one GPU run the slab exchange of csrc/sph_comm.hip (see fake_rccl.hip).  Built in-tree by __graft_entry__.build()
so that it travels to the GPU box; the tests rebuild it if it is missing or older than its source."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.hip")
LIB = os.path.join(HERE, "libfake_rccl.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = f"{LIB}.tmp.{os.getpid()}"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", tmp, SRC, "-lpthread"],
                   check=True, cwd=HERE)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
