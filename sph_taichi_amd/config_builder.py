"""Scene-file reader with the interface of the reference's SimConfig
(/root/reference/config_builder.py:4-37): get_cfg(name, enforce_exist) returns
the value or None; the three block getters return possibly-empty lists."""
from __future__ import annotations

import json


class SimConfig:
    def __init__(self, scene_file_path=None, config: dict | None = None, verbose: bool = False) -> None:
        if config is None:
            with open(scene_file_path, "r") as fh:
                config = json.load(fh)
        self.config = config
        if verbose:
            print(self.config)

    def get_cfg(self, name, enforce_exist=False):
        section = self.config["Configuration"]
        if enforce_exist:
            assert name in section, f"missing configuration key {name!r}"
        return section.get(name)

    def _blocks(self, key):
        return self.config.get(key, [])

    def get_rigid_bodies(self):
        return self._blocks("RigidBodies")

    def get_rigid_blocks(self):
        return self._blocks("RigidBlocks")

    def get_fluid_blocks(self):
        return self._blocks("FluidBlocks")
