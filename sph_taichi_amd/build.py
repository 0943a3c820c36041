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
# The PROFILING build (-DSPH_PROFILE): the same sources with the section-ablation masks of SPH_OPT_DEBUG_ABLATE compiled
# in.  Built on demand (bench.py --ablate / --ablate-mask, SPH_HIP_LIB_VARIANT=profile); the product never loads it.
LIB_PROFILE = os.path.join(_HERE, "libsph_hip_profile.so")


def _fingerprint() -> str:
    """Hash of every source, header and flag: staleness must not depend on file times (a snapshot copied to
    another box keeps contents, not necessarily mtimes)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale(profile: bool = False) -> bool:
    lib = LIB_PROFILE if profile else LIB
    stamp = lib + ".stamp"
    if not os.path.exists(lib) or not os.path.exists(stamp):
        return True
    try:
        return open(stamp).read().strip() != _fingerprint() + (" profile" if profile else "")
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False, profile: bool = False) -> str:
    """Several ranks may get here at once (torch.distributed.run starts one process per GPU): one of them builds,
    under a file lock, into a temporary file that is renamed into place; the others wait and find it fresh."""
    lib = LIB_PROFILE if profile else LIB
    stamp = lib + ".stamp"
    if not (force or stale(profile)):
        return lib
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale(profile):
                tmp = f"{lib}.tmp.{os.getpid()}"
                cmd = [hipcc()] + FLAGS + (["-DSPH_PROFILE"] if profile else []) + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True, cwd=CSRC)
                os.replace(tmp, lib)
                with open(stamp + ".tmp", "w") as fh:
                    fh.write(_fingerprint() + (" profile" if profile else "") + "\n")
                os.replace(stamp + ".tmp", stamp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, profile="--profile" in sys.argv))
