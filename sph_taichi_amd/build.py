"""Build libsph_hip.so (hipcc, gfx950 only) in-tree next to this file."""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["sph_api.hip", "sph_sort.hip", "sph_gather.hip", "sph_integrate.hip", "sph_comm.hip"]
HEADERS = [os.path.join(CSRC, "sph_internal.h"), os.path.join(CSRC, "sph_bricks.h"), os.path.join(_HERE, "..", "include", "sph_hip.h")]
LIB = os.path.join(_HERE, "libsph_hip.so")
# -fno-slp-vectorize: the SLP pass packs the x/y lanes of the distance test into v_pk_*_f32, which on gfx950
# costs extra v_mov + s_nop per pair test (measured: slower than plain v_sub/v_fma)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fno-slp-vectorize"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsph_hip.so cannot be built")
    return exe


STAMP = LIB + ".stamp"


def _fingerprint() -> str:
    """Hash of every source, header and flag: staleness must not depend on file times (a snapshot copied to
    another box keeps contents, not necessarily mtimes)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != _fingerprint()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False) -> str:
    """Several ranks may get here at once (torch.distributed.run starts one process per GPU): one of them builds,
    under a file lock, into a temporary file that is renamed into place; the others wait and find it fresh."""
    if not (force or stale()):
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                tmp = f"{LIB}.tmp.{os.getpid()}"
                cmd = [hipcc()] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True, cwd=CSRC)
                os.replace(tmp, LIB)
                with open(STAMP + ".tmp", "w") as fh:
                    fh.write(_fingerprint() + "\n")
                os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
