"""sph_taichi_amd -- MI355X (gfx950) native WCSPH step behind erizmr/SPH_Taichi's
Python API surface (SimConfig / ParticleSystem / SPHBase / WCSPHSolver.step()).

Host code is Python; every per-step computation runs in hand-written HIP kernels
(sph_taichi_amd/csrc/*.hip) reached through the C ABI of include/sph_hip.h via
ctypes.  There is no CPU fallback: creating a ParticleSystem without the built
library or without a gfx950 GPU raises.
"""
from .config_builder import SimConfig
from .particle_system import ParticleSystem
from .sph_base import SPHBase
from .WCSPH import WCSPHSolver
from .DFSPH import DFSPHSolver

__all__ = ["SimConfig", "ParticleSystem", "SPHBase", "WCSPHSolver", "DFSPHSolver"]
__version__ = "0.1.0"
