"""Worker for the multi-process slab tests (launched once per rank by the tests).

mode "transport": CPU only -- TorchTransport over gloo with ragged fake record buffers.
mode "slabs"    : needs a GPU -- every rank runs a SlabSolver on cuda:0 (gloo staging through the
                  host), rank 0 gathers positions by pid and writes them for the parent to compare.
mode "rccl1"    : needs a GPU -- ONE rank on backend "nccl" (= RCCL on ROCm), device tensors end to end:
                  (1) the transport protocol with the rank as its own left neighbour (count announcement, ragged
                  record payloads, fixed-size swap: RCCL send/recv to self), (2) the 16-sum all-reduce, (3) a
                  SlabSolver with shape-matched bodies and re-cut events stepping over TorchTransport (its
                  all-reduces run through RCCL), positions written for the parent to compare with the single domain.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch
    import torch.distributed as dist
    if mode in ("rccl1", "native1"):
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from sph_taichi_amd.distributed import TorchTransport, NativeTransport, SlabSolver, RECORD_BYTES
    if mode == "native1":
        # The exchange behind the C ABI (csrc/sph_comm.hip) on the hardware there is: ONE rank, its own left neighbour.
        import json
        dev = torch.device("cuda", 0)
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        s = SlabSolver(sd, 0, 1, device=0, recut_every=2, check_every=3)
        tr = NativeTransport(s.ps, dev, loopback=True)
        ok = True
        for it in range(5):                  # ragged payloads, sometimes empty; counts announced one round ahead
            n = (3 + 2 * it) % 5
            payload = (torch.arange(max(n, 1) * RECORD_BYTES, device=dev) % 251 + it).to(torch.uint8)
            torch.cuda.synchronize()
            alloc = lambda from_left, m: torch.zeros(m * RECORD_BYTES, dtype=torch.uint8, device=dev)
            rL, mL, rR, mR = tr.exchange(payload, n, None, 0, alloc)
            s.ps.sync()
            ok &= mL == n and mR == 0 and (n == 0 or bool(torch.equal(rL[: n * RECORD_BYTES], payload[: n * RECORD_BYTES])))
            if it < 4:
                tr.start_counts((3 + 2 * (it + 1)) % 5, 0)
        a = torch.arange(4096, device=dev, dtype=torch.int32).view(torch.uint8)     # DFSPH ghost-velocity refresh path
        b = torch.zeros_like(a)
        torch.cuda.synchronize()
        tr.swap(a, None, b, None)
        s.ps.sync()
        ok &= bool(torch.equal(a, b))
        t = torch.arange(16, dtype=torch.float64, device=dev) * 3.0               # the bodies' 16 shape-matching sums
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t.cpu(), torch.arange(16, dtype=torch.float64) * 3.0))
        ti = torch.tensor([41], dtype=torch.int64, device=dev)                      # the conservation guard's count
        ok &= int(tr.all_reduce_sum(ti).item()) == 41
        ms, nx = tr.halo_time()
        ok &= nx == 5 and ms >= 0.0
        tr.close()
        tr2 = NativeTransport(s.ps, dev)          # a second communicator: the solver's own (world 1, no neighbours)
        s.attach(tr2)
        s.initialize()
        s.step(steps)
        ok &= s.stats.get("recuts", 0) == 0
        o = s.owned(("pid", "x", "v", "density"))
        np.savez(out, ok=np.int32(1 if ok else 0), pid=o["pid"], x=o["x"], v=o["v"], density=o["density"],
                 backend=np.array("native-rccl"))
        tr2.close()
        s.close()
    elif mode == "rccl1":
        import json
        dev = torch.device("cuda", 0)
        assert dist.get_backend() == "nccl"
        tr = TorchTransport(dev, loopback=True)
        ok = True
        for it in range(5):                  # ragged payloads, sometimes empty; counts announced one round ahead
            n = (3 + 2 * it) % 5
            payload = (torch.arange(max(n, 1) * RECORD_BYTES, device=dev) % 251 + it).to(torch.uint8)
            alloc = lambda from_left, m: torch.zeros(m * RECORD_BYTES, dtype=torch.uint8, device=dev)
            rL, mL, rR, mR = tr.exchange(payload, n, None, 0, alloc)
            ok &= mL == n and mR == 0 and (n == 0 or bool(torch.equal(rL[: n * RECORD_BYTES], payload[: n * RECORD_BYTES])))
            if it < 4:
                tr.start_counts((3 + 2 * (it + 1)) % 5, 0)
        a = torch.arange(4096, device=dev, dtype=torch.int32).view(torch.uint8)     # DFSPH ghost-velocity refresh path
        b = torch.zeros_like(a)
        tr.swap(a, None, b, None)
        ok &= bool(torch.equal(a, b))
        t = torch.arange(16, dtype=torch.float64, device=dev) * 3.0               # the bodies' 16 shape-matching sums
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t.cpu(), torch.arange(16, dtype=torch.float64) * 3.0))
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        s = SlabSolver(sd, 0, 1, device=0, recut_every=2, check_every=3)
        s.attach(TorchTransport(dev))
        s.initialize()
        s.step(steps)
        ok &= s.stats.get("recuts", 0) == 0          # one slab: every re-cut event runs its all-reduce and moves nothing
        o = s.owned(("pid", "x", "v", "density"))
        np.savez(out, ok=np.int32(1 if ok else 0), pid=o["pid"], x=o["x"], v=o["v"], density=o["density"],
                 backend=np.array(dist.get_backend()))
        s.close()
    elif mode == "transport":
        tr = TorchTransport("cpu")
        ok = True
        cnt = lambda it: ((3 + rank + it) % 5 if rank > 0 else 0, (7 * rank + 2 * it) % 6 if rank < world - 1 else 0)
        for it in range(5):
            nL, nR = (3 + rank + it) % 5, (7 * rank + 2 * it) % 6          # ragged, sometimes zero
            if it >= 2:         # from the third round on the counts were announced one round ahead
                assert tr._pending is not None
            mk = lambda n, tag: torch.full((max(n, 1) * RECORD_BYTES,), tag, dtype=torch.uint8)
            alloc = lambda from_left, n: torch.zeros(n * RECORD_BYTES, dtype=torch.uint8)
            rL, mL, rR, mR = tr.exchange(mk(nL, 10 + rank) if rank > 0 else None, nL if rank > 0 else 0,
                                         mk(nR, 100 + rank) if rank < world - 1 else None, nR if rank < world - 1 else 0,
                                         alloc)
            if rank > 0:        # my left neighbour sent me its right range
                exp = (7 * (rank - 1) + 2 * it) % 6
                ok &= mL == exp and (exp == 0 or bool((rL[: exp * RECORD_BYTES] == 100 + rank - 1).all()))
            if rank < world - 1:
                exp = (3 + rank + 1 + it) % 5
                ok &= mR == exp and (exp == 0 or bool((rR[: exp * RECORD_BYTES] == 10 + rank + 1).all()))
            if it >= 1 and it < 4:
                tr.start_counts(*cnt(it + 1))
        t = torch.arange(16, dtype=torch.float64) * (rank + 1)      # the bodies' 16 shape-matching sums
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t, torch.arange(16, dtype=torch.float64) * (world * (world + 1) // 2)))
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            open(out, "w").write("ok" if int(flag.item()) == 1 else "fail")
    else:
        import scenes
        import json
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        recut = int(sys.argv[8]) if len(sys.argv) > 8 else 0
        cuts = json.loads(sys.argv[9]) if len(sys.argv) > 9 else None
        s = SlabSolver(sd, rank, world, device=0, recut_every=recut, cuts=cuts)
        s.attach(TorchTransport(torch.device("cuda", 0)))
        s.initialize()
        s.step(steps)
        o = s.owned(("pid", "x", "v", "density"))
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v for k, v in o.items()})
        if rank == 0:
            pid = np.concatenate([g["pid"] for g in gathered])
            np.savez(out, cuts=np.asarray(s.cuts), recuts=s.stats.get("recuts", 0), pid=pid, x=np.concatenate([g["x"] for g in gathered]),
                     v=np.concatenate([g["v"] for g in gathered]),
                     density=np.concatenate([g["density"] for g in gathered]))
        s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
