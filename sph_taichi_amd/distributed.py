"""x-slab decomposition of one WCSPH domain over several GPUs (the
reference is single-device, so this has no counterpart there).

Slabs.  The global cell grid is cut along x (the slowest flatten axis,
particle_system.py:294) at planes chosen from the per-layer particle histogram so
every rank owns about N/world particles.  Rank r owns global cell layers
[X_r, X_r+1); its local grid adds HALO = 2 ghost layers on each side.

One exchange and ONE sort per step.  The sort makes every set of x-layers one
contiguous index range, and a particle moves at most one cell per step, so after
the sweeps of step n (positions advanced, order still that of step n's sort)

    to the left  neighbour:  old local layers [HALO, 2*HALO+1)
    to the right neighbour:  old local layers [nx-2*HALO-1, nx-HALO)

contain every particle that can be a ghost or an immigrant of that neighbour in
step n+1.  They are two contiguous ranges of packed 48-byte records whose offsets
the host already knows (sph_pack_range: plain D2D copies, no pack kernel, no
sync).  The receiver keeps its previously owned range, appends what arrives and
sorts once; ownership is implied by the NEW position (owned <=> local layer in
[HALO, nx-HALO)): a leaver becomes the neighbour's particle and stays behind as
a ghost; strays that end up outside the local grid hash to a virtual cell behind
all real cells and are truncated away.  With a 2-layer halo the
first ghost layer's densities/pressures are computed locally and correctly, so no
second message is needed inside the step; the reaction of the two-way coupling on
a rigid particle is accumulated by its owner from its (ghost-layer-1) fluid
neighbours, so nothing travels back either.  Over xGMI each message is ~HALO+1
cell layers (C4: ~3 MB) to at most two neighbours: latency-, not bandwidth-bound.

DFSPH (simulationMethod 4) across slabs: its Jacobi sweeps change velocities while positions stay put, so after
every sweep (and after predict_velocity) the ghost layers' velocity records are refreshed from their owners -- the
ghost range of a rank is, record for record, the owner's boundary band (same cells, and inside a cell the order by
persistent id that SPH_OPT_SORT_BY_PID makes both sides use), a plain array copy -- and the density error of each iteration is summed over the ranks' owned particles (one 8-byte
all-reduce), so every rank takes the same number of iterations as the single-domain solver.  Fluid and static
solids only.

Transports: `TorchTransport` (torch.distributed P2P; backend "nccl" = RCCL on
ROCm, "gloo" for the CPU tests) and `LocalTransport` (several logical ranks in one
process on one GPU -- how the slab logic is verified against the single-domain run
where only one GPU is available).

Dynamic solids.  Scenes with dynamic RigidBlocks / RigidBodies use HALO = 3: the
moving boundary volumes (sph_base.py:106-113) of ghost layers 1..2 are then
recomputed locally from complete neighbourhoods, exactly as the owner does.
Shape-matched RigidBodies (sph_base.py:200-260) may straddle cuts: after the
advect every rank adds up 16 sums over ITS OWN particles of the body, one
all-reduce (16 doubles) gives every rank the same cm and rotation, and each rank
moves all its local copies; the halo exchange then follows the solve, so for
these scenes it is not hidden behind the interior force sweep.  Every rank
holds the rest positions of all dynamic bodies (a few 10^4 rows, keyed by
persistent id).  Not carried across ranks: `color`, and `x_0` of anything but
dynamic bodies (neither is read by the solver).
"""
from __future__ import annotations

import copy
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _lib, scene as _scene
from .config_builder import SimConfig
from .particle_system import ParticleSystem

HALO = 2            # fluid + static solids; scenes with dynamic solids use HALO_DYNAMIC
HALO_DYNAMIC = 3
RECORD_BYTES = 48


class LocalTransport:
    """Mailbox for P logical ranks driven in lock-step by one process."""

    def __init__(self, world):
        self.world = world
        self.box = {}

    def post(self, src, dst, buf, count):
        self.box[(src, dst)] = (buf, count)

    def take(self, src, dst):
        return self.box.pop((src, dst), (None, 0))


class TorchTransport:
    """torch.distributed point-to-point exchange with the x-neighbours."""

    def __init__(self, device, loopback=False):
        """`loopback` (tests): this rank is its own LEFT neighbour, so every message of the protocol -- the count
        announcement, the record payload, the fixed-size swap -- goes through the backend's send/recv to itself.
        That is how the RCCL path is executed on a box with one GPU."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.cpu_staging = dist.get_backend() == "gloo"
        self.loopback = bool(loopback)

    def _neighbours(self):
        if self.loopback:
            return self.rank, None
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank < self.world - 1 else None
        return left, right

    def start_counts(self, n_left, n_right):
        """Post (non-blocking) the record counts of the NEXT exchange.  The sender knows them as soon as its
        sort is done, a whole sweep phase before the payload exists, so this round trip leaves the critical path."""
        torch, dist = self.torch, self.dist
        left, right = self._neighbours()
        dev = "cpu" if self.cpu_staging else self.device
        out = {left: torch.tensor([n_left], dtype=torch.int64, device=dev),
               right: torch.tensor([n_right], dtype=torch.int64, device=dev)}
        both = torch.zeros(2, dtype=torch.int64, device=dev)     # one buffer: both counts come back in ONE device-to-host copy
        cin = {p: both[i:i + 1] for i, p in enumerate((left, right)) if p is not None}
        self._cin_both = both
        ops = []
        for p in (left, right):
            if p is not None:
                ops.append(dist.P2POp(dist.isend, out[p], p))
                ops.append(dist.P2POp(dist.irecv, cin[p], p))
        works = dist.batch_isend_irecv(ops) if ops else []
        self._pending = (works, out, cin, (n_left, n_right))

    def resolve_counts(self):
        """Wait for the counts posted by start_counts and read them (a device-to-host copy each on RCCL).  Called
        while the host would otherwise sit waiting for the packers, so that it is off the exchange's critical path."""
        if getattr(self, "_pending", None) is None:
            return
        works, _out, cnt_in, announced = self._pending
        self._pending = None
        for w in works:
            w.wait()
        left, right = self._neighbours()
        vals = self._cin_both.tolist() if cnt_in else [0, 0]
        self._resolved = ({p: int(vals[i]) for i, p in enumerate((left, right)) if p in cnt_in}, announced)

    def exchange(self, send_left, n_left, send_right, n_right, alloc):
        """send_* : uint8 device tensors (or None at the domain ends) holding n_* records.
        Returns (recv_left, n, recv_right, n)."""
        torch, dist = self.torch, self.dist
        left, right = self._neighbours()
        # 1) counts: posted ahead by start_counts (and usually read already, see resolve_counts), or exchanged
        #    now (first call)
        if getattr(self, "_resolved", None) is None:
            if getattr(self, "_pending", None) is None:
                self.start_counts(n_left, n_right)
            self.resolve_counts()
        n_in, announced = self._resolved
        self._resolved = None
        if announced != (n_left, n_right):
            raise RuntimeError(f"rank {self.rank}: announced counts {announced} != sent counts {(n_left, n_right)}")
        # 2) payload
        bufs = {}
        ops = []
        for p, sbuf, n in ((left, send_left, n_left), (right, send_right, n_right)):
            if p is None:
                continue
            if n > 0:
                t = sbuf[: n * RECORD_BYTES]
                ops.append(dist.P2POp(dist.isend, t.cpu() if self.cpu_staging else t, p))
            if n_in[p] > 0:
                r = alloc(p == left, n_in[p])
                bufs[p] = r
                rt = torch.empty(n_in[p] * RECORD_BYTES, dtype=torch.uint8) if self.cpu_staging else r[: n_in[p] * RECORD_BYTES]
                bufs[(p, "stage")] = rt
                ops.append(dist.P2POp(dist.irecv, rt, p))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if not self.cpu_staging and torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()   # received bytes are consumed on another stream
        if self.cpu_staging:
            for p in (left, right):
                if p is not None and n_in.get(p, 0) > 0:
                    bufs[p][: n_in[p] * RECORD_BYTES].copy_(bufs[(p, "stage")])
        return (bufs.get(left), n_in.get(left, 0), bufs.get(right), n_in.get(right, 0))

    def swap(self, send_left, send_right, recv_left, recv_right):
        """Fixed-size exchange with the x-neighbours (sizes known on both sides): uint8 device tensors or None."""
        torch, dist = self.torch, self.dist
        left, right = self._neighbours()
        ops, stage = [], []
        for p, sb, rb in ((left, send_left, recv_left), (right, send_right, recv_right)):
            if p is None:
                continue
            if sb is not None and sb.numel() > 0:
                ops.append(dist.P2POp(dist.isend, sb.cpu() if self.cpu_staging else sb, p))
            if rb is not None and rb.numel() > 0:
                t = torch.empty(rb.numel(), dtype=rb.dtype) if self.cpu_staging else rb
                stage.append((rb, t))
                ops.append(dist.P2POp(dist.irecv, t, p))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if not self.cpu_staging and torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()
        if self.cpu_staging:
            for rb, t in stage:
                rb.copy_(t)

    def all_reduce_sum(self, t):
        """In-place sum over ranks of a small device tensor (the 16 shape-matching sums of one body)."""
        if self.cpu_staging:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        else:
            self.dist.all_reduce(t)
        if self.torch.cuda.is_available() and t.is_cuda:
            self.torch.cuda.current_stream().synchronize()   # consumed on the context's stream
        return t


class NativeTransport:
    """The exchange behind the C ABI (csrc/sph_comm.hip): RCCL send / recv enqueued on the context's communication
    stream right behind the halo packers, the next step's counts announced after the sort -- the host never waits for
    the GPU inside a step.  torch.distributed (any backend) is used ONCE, to hand rank 0's 128-byte RCCL unique id to
    the other ranks; `unique_id` (bytes) skips even that (MPI, a shared file, ...).  Same interface as TorchTransport;
    `stream_ordered` tells SlabSolver not to block on the packers' event before calling exchange()."""

    stream_ordered = True

    def __init__(self, ps, device, rank=None, world=None, unique_id=None, loopback=False):
        import torch
        self.torch = torch
        self.ps, self.lib, self.ctx = ps, ps._lib, ps._ctx
        self.device = device
        self.loopback = bool(loopback)
        if rank is None or world is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
        self.rank, self.world = int(rank), int(world)
        if unique_id is None:
            uid = (C.c_uint8 * 128)()
            if self.rank == 0:
                rc = self.lib.sph_comm_unique_id(uid)
                if rc:
                    raise _lib.SphError(f"sph_comm_unique_id: {self.lib.sph_comm_last_error().decode()}")
            if self.world > 1:
                import torch.distributed as dist
                on_dev = dist.get_backend() == "nccl"
                t = torch.tensor(list(uid), dtype=torch.uint8, device=device if on_dev else "cpu")
                dist.broadcast(t, 0)
                uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        else:
            uid = (C.c_uint8 * 128)(*bytes(unique_id))
        comm = C.c_void_p()
        rc = self.lib.sph_comm_create(self.ctx, uid, self.rank, self.world, C.byref(comm))
        _lib.check(self.lib, self.ctx, rc, "sph_comm_create")
        self.comm = comm
        self._announced = None
        self._incoming = None

    def close(self):
        if getattr(self, "comm", None):
            self.lib.sph_comm_destroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _neighbours(self):
        if self.loopback:
            return self.rank, -1
        return (self.rank - 1 if self.rank > 0 else -1), (self.rank + 1 if self.rank < self.world - 1 else -1)

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(self.ctx, self.comm, *args)
        _lib.check(self.lib, self.ctx, rc, name)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None

    def start_counts(self, n_left, n_right):
        left, right = self._neighbours()
        self._call("sph_slab_announce", left, right, int(n_left) if left >= 0 else 0, int(n_right) if right >= 0 else 0)
        self._announced = (int(n_left), int(n_right))
        self._incoming = None

    def resolve_counts(self):
        if self._announced is None or self._incoming is not None:
            return
        a, b = C.c_int32(), C.c_int32()
        self._call("sph_slab_incoming", C.byref(a), C.byref(b))
        self._incoming = (int(a.value), int(b.value))

    def exchange(self, send_left, n_left, send_right, n_right, alloc, after_packers=False):
        left, right = self._neighbours()
        if self._announced is None:
            self.start_counts(n_left, n_right)
        if self._announced != (int(n_left), int(n_right)):
            raise RuntimeError(f"rank {self.rank}: announced counts {self._announced} != sent counts {(n_left, n_right)}")
        self.resolve_counts()
        in_l, in_r = self._incoming
        in_l = in_l if left >= 0 else 0
        in_r = in_r if right >= 0 else 0
        rl = alloc(True, in_l) if in_l > 0 else None
        rr = alloc(False, in_r) if in_r > 0 else None
        self._call("sph_slab_exchange", left, right,
                   self._ptr(send_left) if left >= 0 and n_left > 0 else None, int(n_left) if left >= 0 else 0,
                   self._ptr(send_right) if right >= 0 and n_right > 0 else None, int(n_right) if right >= 0 else 0,
                   self._ptr(rl), in_l, self._ptr(rr), in_r, 1 if after_packers else 0)
        self._announced = None
        self._incoming = None
        return rl, in_l, rr, in_r

    def swap(self, send_left, send_right, recv_left, recv_right):
        left, right = self._neighbours()
        nb = lambda t, ok: int(t.numel() * t.element_size()) if (ok and t is not None) else 0
        self._call("sph_comm_swap", left, right,
                   self._ptr(send_left) if left >= 0 else None, nb(send_left, left >= 0),
                   self._ptr(send_right) if right >= 0 else None, nb(send_right, right >= 0),
                   self._ptr(recv_left) if left >= 0 else None, nb(recv_left, left >= 0),
                   self._ptr(recv_right) if right >= 0 else None, nb(recv_right, right >= 0))

    def all_reduce_sum(self, t):
        """In-place sum over ranks of a small device tensor (float64 or int64); returns after it is complete."""
        torch = self.torch
        if t.dtype not in (torch.float64, torch.int64) or not t.is_cuda:
            raise TypeError("NativeTransport.all_reduce_sum: float64 / int64 device tensors only")
        torch.cuda.current_stream(t.device).synchronize()      # the tensor was filled on torch's stream
        self._call("sph_comm_all_reduce", C.c_void_p(t.data_ptr()), int(t.numel()), 0 if t.dtype == torch.float64 else 1)
        self.ps.sync()                                         # (the main stream waits for the communication stream)
        return t

    def info(self):
        """(rank, world) of the communicator as RCCL reports them (ncclCommUserRank / ncclCommCount)."""
        r, w = C.c_int32(-1), C.c_int32(-1)
        self._call("sph_comm_info", C.byref(r), C.byref(w))
        return int(r.value), int(w.value)

    def halo_time(self):
        ms, n = C.c_double(), C.c_int64()
        self._call("sph_comm_halo_time", C.byref(ms), C.byref(n))
        return float(ms.value), int(n.value)


def negotiate_native_transport(ps, device, create_timeout_s=180.0):
    """COLLECTIVE over torch.distributed's default group: every rank gets a NativeTransport, or every rank gets None
    (and the reason) -- never a mix, and never a rank left alone inside a collective (the fall-back used to be
    collective only if every rank failed the same way).  Each stage ends with a MIN all-reduce of an ok flag, so all
    ranks leave together at the first stage any of them fails:
      0. librccl can be opened here (sph_comm_available) -- the likeliest failure, caught before anything collective;
      1. rank 0 makes the unique id and broadcasts [ok flag | 128 bytes]: a failed id is seen by all;
      2. sph_comm_create (= ncclCommInitRank, itself collective) under a watchdog: a rank whose call does not return
         within `create_timeout_s` votes 0 (its daemon thread is abandoned);
      3. a probe all-reduce of ones must give the world size, under the same watchdog."""
    import threading
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    fdev = device if dist.get_backend() == "nccl" else "cpu"

    def agree(ok):
        f = torch.tensor([1 if ok else 0], dtype=torch.int64, device=fdev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return int(f.item()) == 1

    def watchdog(fn):
        box = {}

        def work():
            try:
                box["value"] = fn()
            except Exception as e:      # noqa: BLE001 -- any failure is a vote for the other transport
                box["error"] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(create_timeout_s)
        if th.is_alive():
            box["error"] = TimeoutError(f"no return within {create_timeout_s:.0f} s")
        return box

    lib = ps._lib
    ok0 = lib.sph_comm_available() == 0
    why0 = "" if ok0 else lib.sph_comm_last_error().decode()
    if not agree(ok0):
        return None, f"stage 0, librccl unavailable on some rank{': ' + why0 if why0 else ''}"
    buf = torch.zeros(129, dtype=torch.uint8, device=fdev)
    if rank == 0:
        uid = (C.c_uint8 * 128)()
        rc = lib.sph_comm_unique_id(uid)
        if rc == 0:
            buf = torch.tensor([1] + list(uid), dtype=torch.uint8, device=fdev)
    dist.broadcast(buf, 0)
    got = buf.cpu().tolist()
    if got[0] != 1:
        return None, "stage 1, rank 0 could not make a unique id"
    uid_bytes = bytes(got[1:])
    box = watchdog(lambda: NativeTransport(ps, device, rank=rank, world=world, unique_id=uid_bytes))
    tr = box.get("value")
    if not agree(tr is not None):
        if tr is not None:
            tr.close()
        return None, f"stage 2, sph_comm_create failed on some rank{': ' + repr(box['error']) if 'error' in box else ''}"

    def probe():
        t = torch.ones(1, dtype=torch.int64, device=device)
        return int(tr.all_reduce_sum(t).item()) == world
    box = watchdog(probe)
    if not agree(box.get("value") is True):
        if "error" not in box or not isinstance(box["error"], TimeoutError):
            tr.close()
        return None, f"stage 3, the probe all-reduce failed on some rank{': ' + repr(box['error']) if 'error' in box else ''}"
    return tr, ""


def plan_recut(cuts, hist, world, halo, width_cap=None):
    """New cut planes, each at most ONE cell layer from the old one, towards the cuts that balance the particle
    counts of `hist` (particles per global x layer).  One layer per event is what the running exchange can absorb:
    the rank that gains a layer already holds it as a ghost, and its neighbour packs one layer deeper for that one
    exchange.  A cut only moves into a slab that is at least halo + 2 layers wide (the deeper pack must come from
    owned layers), no slab drops below halo + 1 layers or grows beyond `width_cap[i]` (its allocation).
    Pure function of its arguments: every rank computes the same answer from the all-reduced histogram."""
    target = _scene.slab_cuts(hist, world, min_width=halo + 1)
    w = [cuts[i + 1] - cuts[i] for i in range(world)]
    d = [0] * (world + 1)
    for i in range(1, world):
        if target[i] > cuts[i] and w[i] >= halo + 2:
            d[i] = 1            # moves right: slab i gives its first layer to slab i-1
        elif target[i] < cuts[i] and w[i - 1] >= halo + 2:
            d[i] = -1
    changed = True
    while changed:
        changed = False
        for i in range(world):
            nw = w[i] + d[i + 1] - d[i]
            if nw < halo + 1:                                   # cancel what shrinks slab i
                if d[i] == 1:
                    d[i], changed = 0, True
                if d[i + 1] == -1:
                    d[i + 1], changed = 0, True
            elif width_cap is not None and nw > width_cap[i]:   # cancel what widens it
                if d[i] == -1:
                    d[i], changed = 0, True
                if d[i + 1] == 1:
                    d[i + 1], changed = 0, True
    return [c + k for c, k in zip(cuts, d)]


class SlabSolver:
    """One rank of the slab-decomposed WCSPH solver."""

    def __init__(self, scene_dict, rank, world, device=0, cuts=None, capacity_factor=1.5, use_torch_stream=False,
                 gather_impl=1, brick_shape=0, scene_dir=None, recut_every=0, nx_slack=16, check_every=64, state=None):
        """`state` = {"x", "v"} by persistent id (e.g. gather_by_pid of an earlier run): the job restarts from there; the
        cuts are planned on the restart positions.
        `recut_every` = K > 0: every K steps the ranks add up their per-layer particle counts and every cut plane
        moves ONE cell layer towards the position that balances the particle counts (`plan_recut`); the slab may
        grow by `nx_slack` layers over its initial width before the allocation is the limit."""
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        cfg = SimConfig(config=copy.deepcopy(scene_dict))
        geom = _scene.Geometry(cfg)
        self.nx_global = int(geom.grid_num[0])
        dyn_blocks = [b for b in cfg.get_rigid_blocks() if b.get("isDynamic")]
        dyn_bodies = [b for b in cfg.get_rigid_bodies() if b.get("isDynamic")]
        self.has_dynamic = bool(dyn_blocks or dyn_bodies)
        self.halo = halo = HALO_DYNAMIC if self.has_dynamic else HALO
        hist = _scene.x_layer_histogram(cfg, base_dir=scene_dir, state=state)
        if cuts is None:
            cuts = _scene.slab_cuts(hist, world, min_width=halo + 1)
        self.cuts = list(cuts)
        self.x_lo, self.x_hi = self.cuts[rank], self.cuts[rank + 1]
        if self.x_hi - self.x_lo < halo + 1:
            raise ValueError(f"slab {rank} is {self.x_hi - self.x_lo} layers wide; need >= {halo + 1}")
        own = int(hist[self.x_lo:self.x_hi].sum())
        per_layer = int(hist.max())
        if int(recut_every) > 0:      # a re-cut balances towards N / world owned particles, whatever the start was
            own = max(own, -(-int(hist.sum()) // world))
        capacity = int(capacity_factor * own) + (2 * halo + 2) * 2 * per_layer + 1024
        self.device = device
        self.tdev = torch.device("cuda", device)
        stream = torch.cuda.current_stream(self.tdev).cuda_stream if use_torch_stream else None
        self.recut_every = int(recut_every)
        self.width_cap = [self.cuts[i + 1] - self.cuts[i] + int(nx_slack) for i in range(world)]   # allocations
        self.steps_done = 0
        # Conservation guard: the exchange carries HALO + 1 layers per side, i.e. it assumes that nothing moves more
        # than one cell layer per step (CFL-sized time steps do).  A particle that does -- a violent WCSPH event, a
        # far too large dt -- lands in the virtual cell and is truncated away, or is kept by two ranks.  Every
        # `check_every` steps (and at every re-cut) the ranks add up their owned counts and raise if the sum is not
        # the scene's particle count, instead of carrying on with a silently different fluid.
        self.n_global = int(hist.sum())
        self.check_every = int(check_every)
        self._shift = (0, 0)        # pending move of (left cut, right cut), applied at the next exchange
        self._recut_due = False
        self._guard_due = False
        self.ps = ParticleSystem(cfg, device=device, stream=stream, scene_dir=scene_dir, state=state,
                                 slab=dict(x_lo=self.x_lo, x_hi=self.x_hi, halo=halo, capacity=capacity,
                                           nx_slack=int(nx_slack) if self.recut_every > 0 else 0))
        self.ps.set_option(_lib.OPT_GATHER_IMPL, gather_impl)
        self.ps.set_option(_lib.OPT_BRICK_SHAPE, brick_shape)
        self.ps.set_option(_lib.OPT_NO_DYNAMIC_SOLIDS, 0 if self.has_dynamic else 1)
        self.ps.set_option(_lib.OPT_SLAB_DROP_OUTSIDE, 1)
        # one-gather force sweep: the host knows whether every fluid block has the same density (=> one particle
        # mass); then the device checks the local particles once and later arrivals are vouched for
        same_mass = len({float(b["density"]) for b in cfg.get_fluid_blocks()}) <= 1
        self.ps.set_option(_lib.OPT_UNIFORM_FLUID, 1 if same_mass else 0)
        # ... and whether the scene has any solid at all: without one every m_V stays m_V0 on every rank, and the density sweep
        # may run its pure-fluid instance (bit-identical; a rank cannot check its arrivals, the scene file can vouch for them)
        if same_mass and self.ps._scene.solid_particle_num == 0:
            self.ps.set_option(_lib.OPT_PURE_FLUID_INSTANCE, 2)
        nxl = self._set_target_layers()
        self.solver = self.ps.build_solver()
        self.dfsph = cfg.get_cfg("simulationMethod") == 4
        if self.dfsph:   # ghost velocities are refreshed record for record: both sides must order a cell the same way
            self.ps.set_option(_lib.OPT_SORT_BY_PID, 1)
        self.nx_local = nxl
        self.capacity = capacity
        nbuf = (halo + 2) * 2 * per_layer * RECORD_BYTES + 4096
        self.send_buf = {side: torch.empty(nbuf, dtype=torch.uint8, device=self.tdev) for side in ("L", "R")}
        self.recv_buf = {side: torch.empty(nbuf, dtype=torch.uint8, device=self.tdev) for side in ("L", "R")}
        self.nbuf = nbuf
        self.owned_range = None     # (first, count) of the owned particles in the current order
        self.off = None
        self._need_density = True
        self.transport = None
        self.has_left, self.has_right = rank > 0, rank < world - 1
        self.stats = {"sent": 0, "received": 0}
        self.host_ms = {"forces_pack": 0.0, "exchange": 0.0, "advance": 0.0, "steps": 0}
        # shape-matched bodies (sph_base.py:247-260 iterates object_id_rigid_body; only dynamic ones move)
        sc = self.ps._scene
        self.dynamic_bodies = sorted(sc.dynamic_rigid_ids)
        self.sums = torch.zeros(16, dtype=torch.float64, device=self.tdev)
        for oid in self.dynamic_bodies:
            body = sc.object_collection[oid]
            rest = np.ascontiguousarray(np.asarray(body["voxelizedPoints"], dtype=np.float32))
            pid = np.ascontiguousarray(body["pidStart"] + np.arange(rest.shape[0], dtype=np.int32), dtype=np.int32)
            self.ps._call("sph_upload_rest_positions", pid.ctypes.data_as(C.c_void_p),
                          rest.ctypes.data_as(C.c_void_p), int(rest.shape[0]))

    # -- helpers ------------------------------------------------------------
    def _set_target_layers(self):
        """targets: density on owned + ghost layer 1 (its rho/p feed the owned forces); forces on owned only,
        or also on ghost layer 1 when dynamic solids exist (their owner accumulates the coupling reaction
        from its ghost fluid neighbours)"""
        halo = self.halo
        nxl = self.x_hi - self.x_lo + 2 * halo
        f_lo, f_hi = (halo - 1, nxl - halo + 1) if self.has_dynamic else (halo, nxl - halo)
        self.ps._call("sph_set_target_layers", halo - 1, nxl - halo + 1, f_lo, f_hi)
        return nxl

    def _offsets(self, layers):
        arr = (C.c_int32 * len(layers))(*layers)
        out = (C.c_int32 * len(layers))()
        self.ps._call("sph_layer_offsets", arr, len(layers), out)
        return list(out)

    def _alloc_recv(self, from_left, n):
        need = n * RECORD_BYTES
        side = "L" if from_left else "R"
        if need > self.recv_buf[side].numel():
            self.recv_buf[side] = self.torch.empty(need, dtype=self.torch.uint8, device=self.tdev)
        return self.recv_buf[side]

    # -- step phases ---------------------------------------------------------------
    # `self.off` holds, for the CURRENT order (the one the last sort produced), the record offsets of the layer
    # boundaries [H, 2*H+1, nx-2*H-1, nx-H]: the ranges to send and the owned range are known on
    # the host.  Per step:
    #   phase_forces : force sweep of the boundary layers, halo packers (records as they will be after this
    #                  step's advect), event; then interior force sweep + in-place advect keep the GPU busy
    #   (exchange)   : starts as soon as the packers are done -> hidden behind the interior force sweep
    #   phase_advance: keep the old owned range, append the neighbours' ranges, sort once (position decides
    #                  ownership, strays fall into the virtual cell), density sweep; read the new offsets back
    def pack_ranges(self):
        """(firstL, nL, firstR, nR) of the next exchange: the H + 1 owned layers next to each neighbour -- its H
        ghost layers plus one layer of migration margin -- and one layer more on a side whose cut plane is about
        to move INTO this slab (the neighbour then owns this slab's first layer and needs ghosts one layer deeper)."""
        o, (sL, sR) = self.off, self._shift
        endL = min(self.off_ext[0], o[3]) if sL > 0 else o[1]
        firstR = max(self.off_ext[1], o[0]) if sR < 0 else o[2]
        return (o[0], (endL - o[0]) if self.has_left else 0, firstR, (o[3] - firstR) if self.has_right else 0)

    def next_counts(self):
        _, nL, _, nR = self.pack_ranges()
        return nL, nR

    def _ensure_send_bufs(self, nL, nR):
        for side, n in (("L", nL), ("R", nR)):
            if n * RECORD_BYTES > self.send_buf[side].numel():
                self.send_buf[side] = self.torch.empty(int(1.25 * n) * RECORD_BYTES, dtype=self.torch.uint8,
                                                       device=self.tdev)

    def init_pack(self):
        """First halo exchange (initialize): sort the initial particles, pack the boundary layers as they are."""
        ps = self.ps
        ps._call("sph_sort")
        self._read_offsets(begin=True)
        fL, nL, fR, nR = self.pack_ranges()
        self._ensure_send_bufs(nL, nR)
        ps._call("sph_slab_pack", fL, nL, C.c_void_p(self.send_buf["L"].data_ptr()),
                 fR, nR, C.c_void_p(self.send_buf["R"].data_ptr()))
        self.stats["sent"] += nL + nR
        return self.send_buf["L"], nL, self.send_buf["R"], nR

    def phase_forces(self, pack=True):
        """pack=False (scenes with shape-matched bodies): all force sweeps + advect, nothing packed yet."""
        ps = self.ps
        H = self.halo
        if self._need_density:
            ps._call("sph_slab_density")
            self._need_density = False
        fL, nL, fR, nR = self.pack_ranges() if pack else (self.off[0], 0, self.off[2], 0)
        self._ensure_send_bufs(nL, nR)
        nx = self.nx_local
        extra = 1 if self.has_dynamic else 0      # coupling reactions on boundary solids come from one layer further in
        sL, sR = self._shift                      # a re-cut in flight: that side packs one layer more
        # a side without a neighbour has no boundary set (its layers are interior: nothing to pack, nothing to hurry)
        bl = (H, 2 * H + 1 + extra + max(sL, 0)) if (self.has_left or not pack) else (H, H)
        br = (nx - 2 * H - 1 - extra - max(-sR, 0), nx - H) if (self.has_right or not pack) else (nx - H, nx - H)
        ps._call("sph_slab_forces", bl[0], bl[1], br[0], br[1],
                 fL, nL, C.c_void_p(self.send_buf["L"].data_ptr()),
                 fR, nR, C.c_void_p(self.send_buf["R"].data_ptr()))
        if getattr(self, "transport", None) is not None and hasattr(self.transport, "resolve_counts"):
            self.transport.resolve_counts()        # the neighbours' counts, read while the packers are still queued
        self._packers_pending = pack and bool(getattr(getattr(self, "transport", None), "stream_ordered", False))
        if not self._packers_pending:
            ps._call("sph_slab_wait_pack")
        self.stats["sent"] += nL + nR
        return self.send_buf["L"], nL, self.send_buf["R"], nR

    def pack_now(self):
        """Pack the boundary layers as they are now (after advect + rigid solve); synchronises."""
        fL, nL, fR, nR = self.pack_ranges()
        self._ensure_send_bufs(nL, nR)
        self._packers_pending = False
        self.ps._call("sph_slab_pack", fL, nL, C.c_void_p(self.send_buf["L"].data_ptr()),
                      fR, nR, C.c_void_p(self.send_buf["R"].data_ptr()))
        self.stats["sent"] += nL + nR
        return self.send_buf["L"], nL, self.send_buf["R"], nR

    def rigid_partial(self, oid, rest=False):
        """This rank's 16 shape-matching sums of body `oid` (device tensor, ready for the all-reduce).  `rest`: the sums
        feed the REST centre of mass (mode 0) -- of a restarted job they are taken over x_0, not over the restart positions
        (in the reference x_0 == x when initialize() computes rigid_rest_cm, sph_base.py:80-90)."""
        first, count = self.owned_range
        from_x0 = rest and getattr(self.ps, "_restarted", False)
        if from_x0:
            self.ps.set_option(_lib.OPT_RIGID_SUMS_FROM_X0, 1)
        try:
            self.ps._call("sph_rigid_partial_sums", int(oid), first, count, C.c_void_p(self.sums.data_ptr()))
        finally:
            if from_x0:
                self.ps.set_option(_lib.OPT_RIGID_SUMS_FROM_X0, 0)
        self.ps.sync()
        return self.sums

    def rigid_apply(self, oid, mode):
        """mode 0: rest centre of mass (sph_base.py:87-89); 1: solve_constraints + enforce_boundary_3D(solid)
        (sph_base.py:247-260)."""
        self.ps._call("sph_rigid_apply_sums", int(oid), C.c_void_p(self.sums.data_ptr()), int(mode))
        if mode == 1:
            self.ps._call("sph_enforce_boundary_3D", _scene.MATERIAL_SOLID)

    def solve_rigid_bodies(self, mode=1):
        for oid in self.dynamic_bodies:
            self.transport.all_reduce_sum(self.rigid_partial(oid, rest=(mode == 0)))
            self.rigid_apply(oid, mode)

    def phase_advance(self, recv_left, n_left, recv_right, n_right, density=True):
        ps = self.ps
        H = self.halo
        o = self.off
        self.stats["received"] += n_left + n_right
        if self._shift != (0, 0):
            # the re-cut takes effect with this sort: the window moves, the kept (old owned) range and the received
            # ranges are re-classified by position -- a lost layer stays behind as a ghost, a gained one arrived
            # with the neighbour's (one layer deeper) pack
            self.x_lo += self._shift[0]
            self.x_hi += self._shift[1]
            self.ps.set_slab_window(self.x_lo, self.x_hi)
            self.nx_local = self._set_target_layers()
            self._shift = (0, 0)
            self.stats["recuts"] = self.stats.get("recuts", 0) + 1
        nl = len(self._layers())
        layers = (C.c_int32 * nl)(*self._layers())
        ps._call("sph_slab_advance", o[0], o[3] - o[0],
                 C.c_void_p(recv_left.data_ptr()) if n_left > 0 else None, n_left,
                 C.c_void_p(recv_right.data_ptr()) if n_right > 0 else None, n_right, layers, nl, 2 if density else 0)
        self._need_density = not density
        self._read_offsets(begin=False)
        # a step that ends with a collective (conservation guard, re-cut) announces AFTER it: no collective may sit
        # between the posted count messages and their payload on the same communicator
        if not (self._recut_due or self._guard_due):
            self.announce()

    def check_conservation(self):
        """Collective: sum of the ranks' owned counts == particles of the scene, else RuntimeError on every rank."""
        t = self.torch.tensor([int(self.owned_range[1])], dtype=self.torch.int64, device=self.tdev)
        total = int(self.transport.all_reduce_sum(t).item())
        if total != self.n_global:
            raise RuntimeError(f"slab decomposition lost particle conservation at step {self.steps_done}: the ranks own "
                               f"{total} particles, the scene has {self.n_global} (rank {self.rank} owns {self.owned_range[1]}). "
                               "A particle crossed more than one cell layer in a step (time step too large for the "
                               f"flow speed?) -- the halo exchange carries {self.halo + 1} layers per side")

    def _begin_step(self):
        """Which collectives end the step that is about to run (decided before its exchange is announced)."""
        self._recut_due = self.recut_every > 0 and (self.steps_done + 1) % self.recut_every == 0
        self._guard_due = self.check_every > 0 and ((self.steps_done + 1) % self.check_every == 0 or self._recut_due)

    def _end_step(self):
        """The deferred collectives, then the announcement of the next exchange's counts."""
        if self._guard_due:
            self.check_conservation()
        if self._recut_due:
            self.recut_now()                  # (announces)
        elif self._guard_due:
            self.announce()
        self._guard_due = False

    def announce(self):
        if getattr(self, "transport", None) is not None and hasattr(self.transport, "start_counts"):
            self.transport.start_counts(*self.next_counts())   # the next exchange's sizes are known now

    # -- re-cut ("re-cut every K steps") ------------------------------------------------------
    def local_histogram(self):
        """Particles per GLOBAL x layer in this rank's owned layers (numpy int64 [nx_global]); synchronises."""
        nx, H = self.nx_local, self.halo
        off = np.asarray(self._offsets(list(range(nx + 1))), dtype=np.int64)
        per_layer = np.diff(off)
        hist = np.zeros(self.nx_global, dtype=np.int64)
        g0 = self.x_lo - H                                     # global layer of local layer 0
        for l in range(H, nx - H):
            if 0 <= g0 + l < self.nx_global:
                hist[g0 + l] = per_layer[l]
        return hist

    def plan_recut(self, hist):
        """Every rank calls this with the same global histogram: new cuts, one layer at most from the old ones."""
        new = plan_recut(self.cuts, hist, self.world, self.halo, width_cap=self.width_cap)
        self._shift = (new[self.rank] - self.cuts[self.rank], new[self.rank + 1] - self.cuts[self.rank + 1])
        self.cuts = list(new)

    def recut_now(self):
        """Collective: histogram all-reduce, plan, then announce the next exchange's (possibly deeper) counts."""
        t = self.torch.from_numpy(self.local_histogram()).to(self.tdev)
        self.transport.all_reduce_sum(t)
        self.plan_recut(t.cpu().numpy())
        self._recut_due = False
        self.announce()

    def _layers(self):
        nx, H = self.nx_local, self.halo
        # [5], [6]: DFSPH velocity bands; [7], [8]: the one-layer-deeper pack ranges of a re-cut
        return [H, 2 * H + 1, nx - 2 * H - 1, nx - H, nx, 2 * H, nx - 2 * H, min(2 * H + 2, nx), max(nx - 2 * H - 2, 0)]

    def _read_offsets(self, begin):
        layers = self._layers()
        nl = len(layers)
        if begin:
            self.ps._call("sph_layer_offsets_begin", (C.c_int32 * nl)(*layers), nl)
        out = (C.c_int32 * nl)()
        self.ps._call("sph_layer_offsets_end", out, nl)
        o = list(out)
        self.ps._call("sph_truncate", o[4])               # drop the virtual cell
        self.off = o[:4]
        self.off_ext = (o[7], o[8])
        self.owned_range = (o[0], o[3] - o[0])
        # DFSPH: ghost ranges and the owned bands that are the neighbours' ghosts (layers [H,2H) / [nx-2H,nx-H))
        self.ghost = {"L": (0, o[0]), "R": (o[3], o[4] - o[3])}
        self.band = {"L": (o[0], max(o[5] - o[0], 0)), "R": (min(o[6], o[3]), max(o[3] - o[6], 0))}

    # -- torch.distributed driver --------------------------------------------
    def attach(self, transport):
        self.transport = transport

    def _exchange(self, sL, nL, sR, nR):
        tr = self.transport
        if getattr(tr, "stream_ordered", False):
            # the exchange is enqueued behind the packers' event on the device: nobody waited for them on the host
            pending, self._packers_pending = getattr(self, "_packers_pending", False), False
            return tr.exchange(sL if self.has_left else None, nL, sR if self.has_right else None, nR, self._alloc_recv,
                               after_packers=pending)
        return tr.exchange(sL if self.has_left else None, nL, sR if self.has_right else None, nR, self._alloc_recv)

    # -- DFSPH across slabs -----------------------------------------------------------------------------
    def velocity_band(self, side):
        """(uint8 tensor, count): the velocity records of my owned band next to `side` = that neighbour's ghosts."""
        first, count = self.band[side]
        buf = self.torch.empty(max(count, 1) * 16, dtype=self.torch.uint8, device=self.tdev)
        self.ps._call("sph_copy_velocity_records", first, count, C.c_void_p(buf.data_ptr()), 0)
        self.ps.sync()
        return buf[: count * 16], count

    def set_ghost_velocities(self, side, buf):
        first, count = self.ghost[side]
        if buf is None or count == 0:
            return
        if buf.numel() != count * 16:
            raise RuntimeError(f"rank {self.rank}: ghost range {side} holds {count} records, the neighbour sent {buf.numel() // 16}")
        self.ps._call("sph_copy_velocity_records", first, count, C.c_void_p(buf.data_ptr()), 1)

    def _dfsph_step_requests(self):
        """One SPHBase.step() with DFSPHSolver.substep() (sph_base.py:263-271, DFSPH.py:400-408) as a generator of
        communication requests: ("halo_v", None) -> refresh the ghosts' velocities; ("sum", x) -> send back the sum
        over ranks; ("records", packed) -> the usual record exchange, send back (rL, nL, rR, nR)."""
        ps, sv = self.ps, self.solver
        call = ps._call
        dt = float(np.float32(sv.dt[None]))     # f32-rounded like the reference's ti.field (and like sph_dfsph_step)
        n_fluid = max(int(ps.fluid_particle_num), 1)
        rho0 = float(sv.density_0)
        first, count = self.owned_range

        def error(offset):
            out = C.c_double()
            call("sph_dfsph_compute_density_error_range", C.c_float(offset), first, count, C.byref(out))
            return float(out.value)

        if self.has_dynamic:
            call("sph_compute_boundary_volume", 1)            # sph_base.py:265 (HALO 3 keeps the ghosts' volumes exact)
        call("sph_dfsph_compute_densities")
        call("sph_dfsph_compute_DFSPH_factor")
        it_v = it_p = 0
        if sv.enable_divergence_solver:                       # DFSPH.py:240-283
            call("sph_dfsph_multiply_time_step", C.c_float(1 / dt))
            call("sph_dfsph_compute_density_change")
            while it_v < 1 or it_v < sv.m_max_iterations_v:
                call("sph_dfsph_divergence_solver_iteration_kernel")
                yield ("halo_v", None)
                call("sph_dfsph_compute_density_change")
                total = yield ("sum", error(0.0))
                avg = float(np.float32(total)) / n_fluid
                if avg <= 1.0 / dt * sv.max_error_V * 0.01 * rho0:
                    break
                it_v += 1
            call("sph_dfsph_multiply_time_step", C.c_float(dt))
        call("sph_dfsph_compute_non_pressure_forces")
        call("sph_dfsph_predict_velocity")
        yield ("halo_v", None)
        call("sph_dfsph_multiply_time_step", C.c_float(1 / (dt * dt)))   # DFSPH.py:324-354
        call("sph_dfsph_compute_density_adv")
        while it_p < 1 or it_p < sv.m_max_iterations:
            call("sph_dfsph_pressure_solve_iteration_kernel")
            yield ("halo_v", None)
            call("sph_dfsph_compute_density_adv")
            total = yield ("sum", error(rho0))
            avg = float(np.float32(total)) / n_fluid
            if avg <= sv.max_error * 0.01 * rho0:
                break
            it_p += 1
        self.dfsph_iterations = (it_v, it_p)
        call("sph_dfsph_advect")
        for oid in self.dynamic_bodies:                       # solve_rigid_body (sph_base.py:247-260) across ranks
            yield ("sum_tensor", self.rigid_partial(oid))
            self.rigid_apply(oid, 1)
        call("sph_enforce_boundary_3D", _scene.MATERIAL_FLUID)
        recv = yield ("records", self.pack_now())
        self.phase_advance(*recv, density=False)

    def _serve(self, gen):
        """Drive a request generator with this rank's transport."""
        tr = self.transport
        try:
            req = next(gen)
            while True:
                kind, arg = req
                if kind == "halo_v":
                    sL, _ = self.velocity_band("L") if self.has_left else (None, 0)
                    sR, _ = self.velocity_band("R") if self.has_right else (None, 0)
                    rL = self.torch.empty(self.ghost["L"][1] * 16, dtype=self.torch.uint8, device=self.tdev) if self.has_left else None
                    rR = self.torch.empty(self.ghost["R"][1] * 16, dtype=self.torch.uint8, device=self.tdev) if self.has_right else None
                    tr.swap(sL, sR, rL, rR)
                    self.set_ghost_velocities("L", rL)
                    self.set_ghost_velocities("R", rR)
                    req = gen.send(None)
                elif kind == "sum":
                    t = self.torch.tensor([arg], dtype=self.torch.float64, device=self.tdev)
                    req = gen.send(float(tr.all_reduce_sum(t).item()))
                elif kind == "sum_tensor":                    # in place: the body's 16 shape-matching sums
                    tr.all_reduce_sum(arg)
                    req = gen.send(None)
                else:
                    req = gen.send(self._exchange(*arg))
        except StopIteration:
            pass

    def step(self, n=1):
        """`self.host_ms` accumulates where the HOST spends a step: enqueueing + waiting for the packers
        ("forces_pack"), inside the exchange ("exchange"), and enqueueing the sort + waiting for its layer offsets
        ("advance").  The GPU keeps working through all three (interior force sweep / density sweep), so these are
        not additive GPU costs; they show whether the exchange stays inside its hiding window."""
        import time
        hm = self.host_ms
        if self.dfsph:
            for _ in range(n):
                self._begin_step()
                self._serve(self._dfsph_step_requests())
                self.steps_done += 1
                self._end_step()
            return
        for _ in range(n):
            self._begin_step()
            t0 = time.perf_counter()
            if self.dynamic_bodies:
                self.phase_forces(pack=False)
                self.solve_rigid_bodies()
                sent = self.pack_now()
            else:
                sent = self.phase_forces()
            t1 = time.perf_counter()
            rL, mL, rR, mR = self._exchange(*sent)
            t2 = time.perf_counter()
            self.phase_advance(rL, mL, rR, mR)
            self.steps_done += 1
            self._end_step()
            t3 = time.perf_counter()
            hm["forces_pack"] += (t1 - t0) * 1e3; hm["exchange"] += (t2 - t1) * 1e3; hm["advance"] += (t3 - t2) * 1e3
            hm["steps"] += 1

    def initialize(self):
        """SPHBase.initialize() (sph_base.py:80-85) for a slab: neighbour structure with halos, then the
        static boundary volumes (ghost layer 1 sees complete neighbourhoods, so owned values are exact)."""
        self.solver._push()
        rL, mL, rR, mR = self._exchange(*self.init_pack())
        self.phase_advance(rL, mL, rR, mR, density=False)
        self.solve_rigid_bodies(mode=0)
        self.ps._call("sph_compute_boundary_volume", 0)
        # once more, so that the ghosts carry their owners' boundary volumes into the first step
        rL, mL, rR, mR = self._exchange(*self.init_pack())
        self.phase_advance(rL, mL, rR, mR, density=False)

    # -- inspection (tests) ----------------------------------------------------
    def owned(self, names=("pid", "x", "v")):
        first, count = self.owned_range
        return {n: getattr(self.ps, n).to_numpy()[first:first + count] for n in names}

    def close(self):
        self.ps.close()


def run_local_slabs(solvers, n_steps, initialize=False):
    """Drive P SlabSolvers that live in ONE process (one GPU) in lock-step; a neighbour's send buffer is read
    directly (device pointer) and the bodies' sums are added on the spot -- what the transports do between ranks."""
    P = len(solvers)

    def swap(sent):
        for r, s in enumerate(solvers):
            s.ps.sync()
        out = []
        for r in range(P):
            rl = (sent[r - 1][2], sent[r - 1][3]) if r > 0 else (None, 0)          # left neighbour's right range
            rr = (sent[r + 1][0], sent[r + 1][1]) if r + 1 < P else (None, 0)
            out.append((rl[0], rl[1], rr[0], rr[1]))
        return out

    def solve_bodies(mode):
        for oid in solvers[0].dynamic_bodies:
            total = sum(s.rigid_partial(oid, rest=(mode == 0)).clone() for s in solvers)
            for s in solvers:
                s.sums.copy_(total)
                s.torch.cuda.current_stream().synchronize()
                s.rigid_apply(oid, mode)

    def exchange_and_advance(sent, density):
        for s, r in zip(solvers, swap(sent)):
            s.phase_advance(*r, density=density)

    def recut_if_due():
        s0 = solvers[0]
        for s in solvers:
            s.steps_done += 1
        owned = sum(int(s.owned_range[1]) for s in solvers)      # the conservation guard of SlabSolver.step
        if owned != s0.n_global:
            raise RuntimeError(f"slab decomposition lost particle conservation at step {s0.steps_done}: the slabs own "
                               f"{owned} particles, the scene has {s0.n_global}")
        if s0.recut_every > 0 and s0.steps_done % s0.recut_every == 0:
            hist = sum(s.local_histogram() for s in solvers)
            for s in solvers:
                s.plan_recut(hist)

    if initialize:
        for s in solvers:
            s.solver._push()
        exchange_and_advance([s.init_pack() for s in solvers], False)
        solve_bodies(0)
        for s in solvers:
            s.ps._call("sph_compute_boundary_volume", 0)
        exchange_and_advance([s.init_pack() for s in solvers], False)
        return
    if solvers[0].dfsph:
        for _ in range(n_steps):
            gens = [s._dfsph_step_requests() for s in solvers]
            reqs = [next(g) for g in gens]
            while reqs is not None:
                kind = reqs[0][0]
                assert all(r[0] == kind for r in reqs), "ranks diverged"
                if kind == "halo_v":
                    bands = [{side: (s.velocity_band(side)[0] if ok else None)
                              for side, ok in (("L", s.has_left), ("R", s.has_right))} for s in solvers]
                    for r, s in enumerate(solvers):
                        s.set_ghost_velocities("L", bands[r - 1]["R"] if r > 0 else None)
                        s.set_ghost_velocities("R", bands[r + 1]["L"] if r + 1 < P else None)
                    answers = [None] * P
                elif kind == "sum":
                    answers = [sum(r[1] for r in reqs)] * P
                elif kind == "sum_tensor":
                    total = sum(r[1].clone() for r in reqs)
                    for s in solvers:
                        s.sums.copy_(total)
                        s.torch.cuda.current_stream().synchronize()
                    answers = [None] * P
                else:
                    answers = swap([r[1] for r in reqs])
                nxt = []
                for g, a in zip(gens, answers):
                    try:
                        nxt.append(g.send(a))
                    except StopIteration:
                        nxt.append(None)
                reqs = None if all(x is None for x in nxt) else nxt
            recut_if_due()
        return
    for _ in range(n_steps):
        if solvers[0].dynamic_bodies:
            for s in solvers:
                s.phase_forces(pack=False)
            solve_bodies(1)
            sent = [s.pack_now() for s in solvers]
        else:
            sent = [s.phase_forces() for s in solvers]
        exchange_and_advance(sent, True)
        recut_if_due()


def gather_by_pid(solvers, name, n_global):
    """Global array `name` indexed by persistent id, assembled from every rank's owned particles."""
    out = None
    for s in solvers:
        o = s.owned(("pid", name))
        if out is None:
            shape = (n_global,) + o[name].shape[1:]
            out = np.full(shape, np.nan, dtype=o[name].dtype) if o[name].dtype.kind == "f" else np.full(shape, -1, o[name].dtype)
        out[o["pid"]] = o[name]
    return out


# ---------------------------------------------------------------------------
# bench.py --gpus N
#   default       : weak scaling -- every rank owns one ~1.74 M-particle slab of the tiled C3' box (`value`), and, in the same
#                   run, BASELINE.json's config 5 in its own geometry (`c4_dambreak`: 13.9 M particles, travelling cuts)
#   --workload c4_dambreak : that dam-break as the line itself (strong scaling: the job is 13,939,200 particles at any N)
# ---------------------------------------------------------------------------
_BENCH_CFG = {
    "domainStart": [0.0, 0.0, 0.0], "particleRadius": 0.01, "numberOfStepsPerRenderUpdate": 1, "density0": 1000,
    "simulationMethod": 0, "gravitation": [0.0, -9.81, 0.0], "timeStepSize": 0.0004, "stiffness": 50000, "exponent": 7,
    "boundaryHandlingMethod": 0, "exportFrame": False, "exportPly": False, "exportObj": False,
}


def _box_scene(domain_end, counts, corner=(0.04, 0.04, 0.04)):
    d = 0.02
    end = [c + (n - 0.5) * d for c, n in zip(corner, counts)]
    return {"Configuration": dict(_BENCH_CFG, domainEnd=list(domain_end)),
            "FluidBlocks": [{"objectId": 0, "start": list(corner), "end": end, "translation": [0.0, 0.0, 0.0],
                             "scale": [1, 1, 1], "velocity": [0.0, 0.0, 0.0], "density": 1000.0,
                             "color": [50, 100, 200]}]}, counts[0] * counts[1] * counts[2]


def slab_bench_scene(world):
    """Weak-scaling family: `world` copies of BASELINE.md's C3' box side by side along x --
    (246*world) x 74 x 96 particles in a (5*world, 3, 2) tank, so every rank's slab is exactly the N = 1
    workload (1,747,584 particles, 125 x 75 x 50 cells) plus its halos; world = 8 gives 13,980,672 particles
    (BASELINE.json: "13.9 M particles" on 8 GPUs)."""
    return _box_scene([5.0 * world, 3.0, 2.0], (246 * world, 74, 96))


def c4_dambreak_scene(scale=1.0):
    """BASELINE.json config 5 / SURVEY 8(d) C4 in its own geometry: the (16, 4, 3.4) tank (400 x 100 x 85 cells) with
    the 512 x 165 x 165 = 13,939,200-particle column at its -x end.  The column collapses into the empty 5.8 m of the
    tank, so the cuts that balance the ranks travel: the case `recut_every` exists for.  `scale` < 1 shrinks the
    column's particle counts (tests, smoke runs)."""
    counts = tuple(max(int(round(c * scale)), 8) for c in (512, 165, 165))
    return _box_scene([16.0, 4.0, 3.4], counts)


def _choose_transport(s, rank, local_rank):
    """The exchange: RCCL behind the C ABI (NativeTransport: enqueued on the device, no host wait inside a step) when the
    job runs on RCCL -- agreed on collectively, stage by stage (negotiate_native_transport) --; torch.distributed P2P
    otherwise (gloo: several ranks sharing one GPU) or on request (SPH_TRANSPORT=torch)."""
    import torch
    import torch.distributed as dist
    # ("nccl", or the mixed "cpu:gloo,cuda:nccl": control plane over gloo, RCCL for everything on the device)
    want = os.environ.get("SPH_TRANSPORT", "native" if "nccl" in str(dist.get_backend()) else "torch")
    transport = None
    if want == "native":
        transport, why = negotiate_native_transport(s.ps, torch.device("cuda", local_rank))
        if transport is None and rank == 0:
            print(f"[bench] native RCCL transport not used ({why}); every rank falls back to torch.distributed P2P",
                  file=sys.stderr, flush=True)
    if transport is None:
        transport = TorchTransport(torch.device("cuda", local_rank))
    return transport


def _timed_steps(s, steps, red_dev):
    """Exactly `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks (seconds)."""
    import time
    import torch
    import torch.distributed as dist
    s.ps.sync()
    torch.cuda.synchronize()
    _barrier(red_dev)
    t0 = time.perf_counter()
    s.step(steps)
    s.ps.sync()
    torch.cuda.synchronize()
    _barrier(red_dev)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt.item())


def _barrier(ctl_dev):
    """A barrier on the job's CONTROL device: an all-reduce of one element there.  (dist.barrier() picks a device itself; with
    the mixed backend "cpu:gloo,cuda:nccl" it would create the RCCL process group that this configuration exists to avoid.)"""
    import torch
    import torch.distributed as dist
    t = torch.zeros(1, dtype=torch.int32, device=ctl_dev)
    dist.all_reduce(t)
    if ctl_dev.type == "cuda":
        torch.cuda.synchronize(ctl_dev)


def _phase_events(s, steps):
    """Per-phase HIP events of rank 0 in EXTRA steps outside the timed region: in slab mode (two streams) the timestamping
    barriers cost ~7 % of a step, so they stay out of the number that is reported."""
    tm = _lib.SphTimings()
    s.ps.set_option(_lib.OPT_TIMING, 1)
    s.ps._call("sph_reset_timings")
    s.step(steps)
    s.ps.sync()
    s.ps._call("sph_get_timings", tm)
    s.ps.set_option(_lib.OPT_TIMING, 0)
    kt = max(int(tm.steps), 1)
    return {"sort": round(tm.sort_ms / kt, 4), "neighbour": round(tm.neighbour_ms / kt, 4), "force": round(tm.force_ms / kt, 4),
            "integrate": round(tm.integrate_ms / kt, 4), "sum_of_phases": round(tm.total_ms / kt, 4)}


def _all_owned(s, red_dev):
    import torch
    import torch.distributed as dist
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=red_dev)
    t[dist.get_rank()] = int(s.owned_range[1])
    dist.all_reduce(t)
    return [int(v) for v in t.tolist()]


class _NoWatchdog:
    """run_slab_bench / run_c4_dambreak outside bench.py (tests, tools): stage names go nowhere."""

    def stage(self, name, budget_s=None):
        pass

    def keep(self, line, key):
        pass

    def remaining(self):
        return float("inf")


class CollectiveStageError(RuntimeError):
    """Raised on EVERY rank when any rank failed a stage of a collectively guarded sequence."""


def _ctl_device(local_rank):
    """Where the job's small control tensors (ok flags, timings, owned counts) live: on the GPU under RCCL, on the host under
    gloo (several ranks sharing one GPU)."""
    import torch
    import torch.distributed as dist
    # ("cpu:gloo,cuda:nccl": the control plane stays on the host, and torch's RCCL process group is never created unless the
    # torch transport has to carry the records)
    return torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")


def collective_stage(name, fn, ctl_dev, wd=None):
    """Run this rank's share `fn` of stage `name`, then agree: ONE MAX all-reduce of (rank + 1 if it raised, else 0).  Either
    every rank returns fn's value or every rank raises CollectiveStageError naming the stage and the (highest) failing
    rank -- never one rank in an except-branch while its peers wait for it in the next stage's collectives (ADVICE r04
    medium).  A failure INSIDE a stage that itself contains collectives can still leave the peers waiting; that is what
    the bench's wall-clock watchdog is for."""
    import sys
    import torch
    import torch.distributed as dist
    if wd is not None:
        wd.stage(name)
    err, val = None, None
    try:
        val = fn()
    except Exception as e:      # noqa: BLE001 -- any failure is reported to all
        err = f"{type(e).__name__}: {e}"
        print(f"[bench] rank {dist.get_rank()}: stage '{name}' failed: {err}", file=sys.stderr, flush=True)
    f = torch.tensor([0 if err is None else dist.get_rank() + 1], dtype=torch.int64, device=ctl_dev)
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    bad = int(f.item())
    if bad:
        raise CollectiveStageError(f"stage '{name}' failed on rank {bad - 1}" + (f" (this rank: {err})" if err else ""))
    return val


def run_c4_dambreak(args, rank, world, local_rank, scale=1.0, wd=None):
    """BASELINE.json config 5 across the ranks of this job: 13,939,200 particles cut by particle count, one exchange and
    one sort per step, the cuts re-planned every `--recut-every` (default 10) steps.  Two states: the first collapse
    (W untimed + K timed steps from rest) and `settled` = after `--settled-after` further steps (the front has crossed the
    tank).  Per-rank owned counts and the cuts at both ends say how well the travelling cuts keep the ranks balanced;
    `conserved` = the ranks' owned counts still add up to the scene."""
    import torch
    import torch.distributed as dist
    from .benchutil import REF_PARTICLES
    wd = wd or _NoWatchdog()
    sd, n_global = c4_dambreak_scene(scale)
    recut = int(getattr(args, "recut_every", 0)) or 10
    red_dev = _ctl_device(local_rank)
    box = {}

    def stage(name, fn):
        return collective_stage("c4_dambreak: " + name, fn, red_dev, wd)

    def build():
        box["s"] = SlabSolver(sd, rank, world, device=local_rank, gather_impl=args.gather_impl, brick_shape=args.brick_shape,
                              recut_every=recut, check_every=0)

    def attach():
        s = box["s"]
        box["t"] = _choose_transport(s, rank, local_rank)
        s.attach(box["t"])
        s.initialize()

    try:
        stage("build the slab contexts", build)
        stage("transport + initialize", attach)
        s, transport = box["s"], box["t"]
        out = {"workload": "c4_dambreak_512x165x165_in_16x4x3.4", "particles": n_global, "recut_every": recut,
               "transport": type(transport).__name__, "comm": _comm_info(transport), "halo_layers": s.halo,
               "cuts_start": list(s.cuts), "owned_start": _all_owned(s, red_dev)}

        def warm():
            s.step(args.warmup)
            s.host_ms = {k: 0 if k == "steps" else 0.0 for k in s.host_ms}
        stage("warm-up steps", warm)
        dt = stage("timed steps from rest", lambda: _timed_steps(s, args.steps, red_dev))
        ev = stage("phase events", lambda: _phase_events(s, min(args.steps, 20)))
        out["from_rest"] = {"value": round(args.steps / dt * n_global / REF_PARTICLES, 3), "ms_per_step": round(dt / args.steps * 1e3, 4),
                            "steps": args.steps, "warmup": args.warmup, "breakdown_ms": dict(ev, rank=0)}
        settle = int(getattr(args, "settled_after", 0))
        if settle > 0:
            stage(f"{settle} settling steps", lambda: s.step(settle))
            k = max(args.steps, 50)
            dt = stage("timed steps, settled", lambda: _timed_steps(s, k, red_dev))
            owned = _all_owned(s, red_dev)
            ev = stage("phase events, settled", lambda: _phase_events(s, 20))
            out["settled"] = {"value": round(k / dt * n_global / REF_PARTICLES, 3), "ms_per_step": round(dt / k * 1e3, 4), "steps": k,
                              "after_steps": s.steps_done - k - 20, "breakdown_ms": dict(ev, rank=0),
                              "owned": owned, "imbalance": round(max(owned) / (n_global / world), 4), "cuts": list(s.cuts)}
        owned = _all_owned(s, red_dev)
        out["owned_end"] = owned
        out["cuts_end"] = list(s.cuts)
        out["recut_events_rank0"] = int(s.stats.get("recuts", 0))
        out["conserved"] = sum(owned) == n_global
        out["imbalance_start"] = round(max(out["owned_start"]) / (n_global / world), 4)
        out["imbalance_end"] = round(max(owned) / (n_global / world), 4)
        out["rank0_host_ms_per_step"] = {k_: round(v / max(s.host_ms["steps"], 1), 4) for k_, v in s.host_ms.items() if k_ != "steps"}
        if isinstance(transport, NativeTransport):
            ms, n = transport.halo_time()
            out["halo_device_ms"] = round(ms / max(n, 1), 4)
        st = _lib.SphStats()
        s.ps._call("sph_get_stats", st)
        out["rank0_neighbourhood"] = {"max_list_entries": st.max_list, "list_overflow_targets": st.list_overflow_targets,
                                      "lds_overflow_targets": st.lds_overflow_targets, "max_cell_occupancy": st.max_cell_occupancy}
        return out
    finally:
        wd.stage("c4_dambreak: teardown")
        # (a close() that raises must not mask the stage error that brought us here: ADVICE r05)
        t = box.get("t")
        for obj in ((t if isinstance(t, NativeTransport) else None), box.get("s")):
            if obj is None:
                continue
            try:
                obj.close()
            except Exception as e:      # noqa: BLE001
                print(f"[bench] rank {rank}: c4_dambreak teardown: {type(e).__name__}: {e}", file=sys.stderr, flush=True)


def _comm_info(transport):
    """What the communication library itself reports about the communicator the records travel on."""
    import torch.distributed as dist
    if isinstance(transport, NativeTransport):
        r, w = transport.info()
        return {"library": "RCCL behind the C ABI (sph_comm.hip: ncclCommInitRank)", "world": w, "rank": r}
    return {"library": f"torch.distributed ({dist.get_backend()}) P2P", "world": dist.get_world_size(), "rank": dist.get_rank()}


def run_slab_bench(args, rank, world, local_rank, wd=None):
    """`bench.py --gpus N` on every rank.  `wd` (benchutil.Watchdog) is told what the rank is doing, so that a hang is
    reported with its stage; the tiled line is handed to it (keep) BEFORE the supplementary c4_dambreak object starts."""
    import torch
    import torch.distributed as dist
    from .benchutil import gpu_preheat, _HEAT, REF_PARTICLES, HBM_PEAK_GBS
    wd = wd or _NoWatchdog()
    dfsph = getattr(args, "solver", "wcsph") == "dfsph"
    c4_line = getattr(args, "workload", "") == "c4_dambreak"
    red_dev = _ctl_device(local_rank)
    metric = ("DFSPH steps/sec at 1.74 M particles (supplementary)" if dfsph else
              "WCSPH steps/sec at 1.74 M particles (+ ms/step breakdown sort/neighbour/force)")
    if c4_line and not dfsph:
        # the named 8-GPU workload as the line itself
        c4 = run_c4_dambreak(args, rank, world, local_rank, scale=float(os.environ.get("SPH_C4_SCALE", "1.0")), wd=wd)
        fr = c4["from_rest"]
        return {"metric": metric, "value": fr["value"], "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": fr["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "preheat_ms": 0.0,
                "config": {"workload": c4["workload"], "particles": c4["particles"], "recut_every": c4["recut_every"],
                           "backend": dist.get_backend(), "transport": c4["transport"], "comm": c4["comm"],
                           "particles_owned_per_rank": c4["owned_end"],
                           "parallelism": f"x-slab x{world} cut by particle count, travelling cuts, 1 exchange/step"},
                "steps_per_s_job": round(fr["value"] * REF_PARTICLES / c4["particles"], 3),
                "breakdown_ms": dict(fr["breakdown_ms"], halo_device=c4.get("halo_device_ms")),
                "settled": c4.get("settled"), "c4_dambreak": c4, "roofline": None, "cpu_baseline": None}
    sd, n_global = slab_bench_scene(world)
    if dfsph:                                     # supplementary line, like bench.py --solver dfsph at N = 1
        sd["Configuration"]["simulationMethod"] = 4
        sd["Configuration"]["timeStepSize"] = 0.004
    # (check_every = 0: the conservation guard is a blocking all-reduce + host read; the bench scene is balanced and
    # slow -- no particle can outrun the halo -- so the guard stays out of the timed region)
    wd.stage("tiled: build the slab contexts")
    s = SlabSolver(sd, rank, world, device=local_rank, gather_impl=args.gather_impl, brick_shape=args.brick_shape,
                   recut_every=getattr(args, "recut_every", 0), check_every=0)
    wd.stage("tiled: transport negotiation")
    transport = _choose_transport(s, rank, local_rank)
    s.attach(transport)
    comm = _comm_info(transport)
    wd.stage("tiled: initialize (first exchange + sort)")
    s.initialize()
    gpu_preheat(local_rank, float(getattr(args, "preheat_ms", 0.0)))    # clock ramp after the host-side set-up: see gpu_preheat
    wd.stage("tiled: warm-up steps")
    s.step(args.warmup)
    s.host_ms = {k: 0 if k == "steps" else 0.0 for k in s.host_ms}
    wd.stage("tiled: timed steps")
    dt = _timed_steps(s, args.steps, red_dev)
    wd.stage("tiled: owned counts + phase events")
    owned = _all_owned(s, red_dev)
    host_ms = dict(s.host_ms)
    phases = {"sort": 0.0, "neighbour": 0.0, "force": 0.0, "integrate": 0.0, "sum_of_phases": 0.0}
    if not dfsph:
        phases = _phase_events(s, min(max(args.steps, 1), 20))
    roofline = None
    if not dfsph and phases["neighbour"] > 0 and phases["force"] > 0:
        # rank 0's dominant sweep over its local records (owned + ghosts) and local cells, algorithmic bytes as at
        # N = 1 (density+EOS 32 N + 4 G, fused force 60 N + 4 G); in slab mode the force phase is an
        # interior launch plus a boundary launch on a side stream, so its figure is per phase, not per launch
        n_loc = s.ps.count()
        g_loc = int(np.prod(s.ps._local_grid_num))
        cands = {"k_gather_brick<GM_DENSITY_EOS>": (32.0 * n_loc + 4.0 * g_loc, phases["neighbour"]),
                 "k_gather_brick<GM_FORCE_FUSED*> (interior + boundary launches)": (60.0 * n_loc + 4.0 * g_loc, phases["force"])}
        dom = max(cands, key=lambda k_: cands[k_][1])
        ab, ms = cands[dom]
        ach = ab / (ms * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "rank": 0, "alg_bytes_per_launch": ab,
                    "avg_launch_ms": round(ms, 4), "local_records": n_loc, "local_cells": g_loc,
                    "note": "rank 0; gather sweeps are VALU-issue-bound, not HBM-bound (DESIGN.md section 4)"}
    steps_per_s = args.steps / dt
    line = {
        "metric": metric,
        "value": round(steps_per_s * n_global / REF_PARTICLES, 3), "unit": "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "preheat_ms": 0.0 if _HEAT.get("broken") else float(getattr(args, "preheat_ms", 0.0)),
        "config": {"workload": f"c3p_tiled_x{world}_{246 * world}x74x96", "particles": n_global,
                   "particles_owned_sum": sum(owned), "particles_owned_per_rank": owned, "cuts": s.cuts, "halo_layers": s.halo,
                   "sent_records_per_step": round(s.stats["sent"] / max(args.steps + args.warmup + 1, 1), 1),
                   "backend": dist.get_backend(), "transport": type(transport).__name__, "comm": comm,
                   "recut_every": s.recut_every,
                   "rank0_host_ms_per_step": {k: round(v / max(host_ms["steps"], 1), 4)
                                              for k, v in host_ms.items() if k != "steps"},
                   "parallelism": f"x-slab x{world}, 1 exchange/step over "
                                  f"{'RCCL' if 'nccl' in str(dist.get_backend()) else dist.get_backend()} P2P"},
        "steps_per_s_job": round(steps_per_s, 3),
        "breakdown_ms": dict(phases, rank=0, halo=round(host_ms["exchange"] / max(host_ms["steps"], 1), 4),
                             note="sort / neighbour / force / integrate: HIP events on rank 0's stream over extra steps "
                                  "after the timed region (sum_of_phases adds these four).  halo: rank 0's wall time "
                                  "inside the record exchange call per step of the timed region (native transport: the enqueue "
                                  "only; torch transport: batch_isend_irecv + wait); it runs beside the interior force sweep, "
                                  "so it is not an additive GPU cost unless it exceeds that sweep"),
        "roofline": roofline, "cpu_baseline": None,
    }
    if dfsph:
        line["config"]["solver"] = "dfsph"
        line["config"]["last_step_iterations"] = list(getattr(s, "dfsph_iterations", (0, 0)))
    if isinstance(transport, NativeTransport):
        ms, n = transport.halo_time()
        line["breakdown_ms"]["halo_device"] = round(ms / max(n, 1), 4)
        line["breakdown_ms"]["note"] += ("; halo_device: mean duration of the payload exchange on rank 0's communication "
                                         "stream (HIP events around ncclGroupStart..End)")
    wd.stage("tiled: teardown")
    if isinstance(transport, NativeTransport):
        transport.close()
    s.close()
    # ... and, in the same driver command, BASELINE.json's named 8-GPU workload in its own geometry:
    # 13.9 M particles whatever N is (strong scaling), unbalanced start, cuts re-planned every 10 steps, two states.
    # It is SUPPLEMENTARY: the tiled line above is complete and is handed to the watchdog first, so that nothing this
    # object does -- an exception on one rank (agreed on collectively, stage by stage: collective_stage), a hang (its own
    # wall-clock budget) -- can cost the contract line.
    if not dfsph and int(getattr(args, "c4", 1)):
        line["c4_dambreak"] = None
        wd.keep(line, "c4_dambreak")
        budget = min(float(getattr(args, "c4_budget_s", 300.0)), max(wd.remaining() - 20.0, 1.0))
        t_end = time.monotonic() + budget

        class _Budgeted:            # every stage of the object inherits what is left of the object's budget
            def stage(self, name, budget_s=None):
                wd.stage(name, budget_s=max(t_end - time.monotonic(), 0.05) if budget != float("inf") else None)

            def keep(self, line, key):
                pass

            def remaining(self):
                return wd.remaining()
        try:
            line["c4_dambreak"] = run_c4_dambreak(args, rank, world, local_rank, scale=float(os.environ.get("SPH_C4_SCALE", "1.0")),
                                                  wd=_Budgeted())
            line["c4_dambreak"]["budget_s"] = round(budget, 1) if budget != float("inf") else None
        except Exception as e:      # noqa: BLE001 -- CollectiveStageError on every rank alike, or a failure outside the guarded stages
            line["c4_dambreak"] = {"error": f"{type(e).__name__}: {e}"}
        wd.stage("c4_dambreak: done")
    return line
