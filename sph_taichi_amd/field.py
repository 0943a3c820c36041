"""Minimal stand-ins for the Taichi field objects the reference exposes.

`DeviceField` is a view of one per-particle array that lives inside the HIP
context (ti.field / ti.Vector.field of particle_system.py:101-113): it supports
the calls the reference's host code makes -- to_numpy(), from_numpy(), fill(),
.shape, and (slow, debugging only) integer indexing.  `HostScalar` replaces a
0-d ti.field such as `particle_num` / `dt` (read and written as `f[None]`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class HostScalar:
    def __init__(self, value=0, dtype=float, on_set=None):
        self._dtype = dtype
        self._v = dtype(value)
        self._on_set = on_set
        self.shape = ()

    def __getitem__(self, idx):
        if idx is not None and idx != ():
            raise IndexError("0-d field: index with [None]")
        return self._v

    def __setitem__(self, idx, value):
        if idx is not None and idx != ():
            raise IndexError("0-d field: index with [None]")
        self._v = self._dtype(value)
        if self._on_set:
            self._on_set(self._v)

    def to_numpy(self):
        return np.asarray(self._v)


class DeviceField:
    def __init__(self, owner, field_id: int, dtype, n_getter, vec: int = 0, writable: bool = True, name: str = ""):
        self._owner = owner          # object with ._lib and ._ctx
        self._fid = field_id
        self.dtype = np.dtype(dtype)
        self._n = n_getter
        self._vec = vec
        self._writable = writable
        self.name = name

    @property
    def shape(self):
        return (self._n(),)

    def _full_shape(self):
        return (self._n(), self._vec) if self._vec else (self._n(),)

    def to_numpy(self):
        out = np.empty(self._full_shape(), dtype=self.dtype)
        lib, ctx = self._owner._lib, self._owner._ctx
        rc = lib.sph_download(ctx, self._fid, out.ctypes.data_as(C.c_void_p), out.nbytes)
        _lib.check(lib, ctx, rc, f"download({self.name})")
        return out

    def from_numpy(self, arr):
        if not self._writable:
            raise _lib.SphError(f"field {self.name} is read-only")
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        if a.shape != self._full_shape():
            raise ValueError(f"{self.name}: expected shape {self._full_shape()}, got {a.shape}")
        lib, ctx = self._owner._lib, self._owner._ctx
        rc = lib.sph_upload(ctx, self._fid, a.ctypes.data_as(C.c_void_p), a.nbytes)
        _lib.check(lib, ctx, rc, f"upload({self.name})")

    def fill(self, value):
        self.from_numpy(np.full(self._full_shape(), value, dtype=self.dtype))

    def __getitem__(self, idx):
        # debugging convenience only: a full download per call
        return self.to_numpy()[idx]

    def __len__(self):
        return self._n()
