"""Import stub so /root/reference/particle_system.py (`import trimesh as tm`) can
be imported under the taichi shim.  The golden scenes use FluidBlocks/RigidBlocks
only, so no trimesh function is ever called; any use raises."""


def __getattr__(name):
    raise RuntimeError(f"trimesh.{name}: trimesh is not available in this image (shim stub)")
